import os, sys, resource, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import model_based_rl_amd
from model_based_rl_amd import train
moves = int(sys.argv[1])
t = time.time()
thr = train.main(['--environment', 'LunarLander-v2', '--num_envs', '4096', '--num_simulations', '30', '--seed', '0', '--selfplay_only',
                  '--max_moves', str(moves), '--prime_moves', '768', '--window_size', '2097152', '--weight_sync_frequency', '128',
                  '--runs_dir', '/tmp/mz_runs'])
print('moves', moves, 'wall %.1f s' % (time.time() - t), 'maxrss MB', resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024,
      'threads', len(os.listdir('/proc/self/task')), 'fds', len(os.listdir('/proc/self/fd')))
