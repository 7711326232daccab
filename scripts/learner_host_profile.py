"""development aid: where the host time of one learner update goes (cProfile of the learner's thread over 1000 updates with
the prefetch thread running).  usage: learner_host_profile.py [no_prefetch]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib.util
spec = importlib.util.spec_from_file_location('ls', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'learner_graph_speed.py'))
ls = importlib.util.module_from_spec(spec)
if len(sys.argv) > 1 and sys.argv[1] == 'no_prefetch':
  os.environ['MZ_LS_NO_PREFETCH'] = '1'
spec.loader.exec_module(ls)
cfg, storage, replay, learner = ls.setup([])
ls.loop(learner, replay, 50)
pr = cProfile.Profile()
pr.enable()
r = ls.loop(learner, replay, 1000)
pr.disable()
print(r)
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
