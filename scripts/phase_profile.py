import sys, types, numpy as np, torch
sys.path.insert(0,'.')
from model_based_rl_amd.engine import Engine, flatten_weights
from model_based_rl_amd.networks import FCNetwork
torch.manual_seed(0)
O,A,SIMS=(int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])) if len(sys.argv)>3 else (8,4,30)
net=FCNetwork(O,A,torch.device('cpu'),types.SimpleNamespace()).eval()
eng=Engine(4096,O,A,SIMS,seed=1)
eng.set_weights(net.state_dict())
obs=torch.randn(4096,O,device='cuda')
names=['gather','bar','dyn_fc1','dyn_fc2','comb1','ln/rew','pred_fc1','pred_fc2','comb2','val/lg','t_expand','t_backup','t_select','t_rest']
for it in range(3):
    eng.initial_inference(obs); eng.root_prepare(None,None,None,device_rng=True,move=it)
    c=eng.search_phase_profile()
print('cycles per sim (avg over WGs), per wave:')
for p in range(14):
    print('%-9s'%names[p], ' '.join('%8.0f'%(c[w,p]/SIMS) for w in range(4)))
print('total    ', ' '.join('%8.0f'%(c[w].sum()/SIMS) for w in range(4)))
