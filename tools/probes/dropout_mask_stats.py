import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, moka_amd.functional as F
dev=torch.device('cuda')
for p in (0.05, 0.1, 0.5):
    m=F.dropout_mask(p, 1234, 4096, 4096, dev).float()
    print(p, 'keep', m.mean().item(), 'col spread', m.mean(0).std().item(), 'row spread', m.mean(1).std().item(),
          'adjacent corr', torch.corrcoef(torch.stack([m[:, :-1].flatten(), m[:, 1:].flatten()]))[0,1].item(),
          'stride8 corr', torch.corrcoef(torch.stack([m[:, :-8].flatten(), m[:, 8:].flatten()]))[0,1].item(),
          'row corr', torch.corrcoef(torch.stack([m[:-1].flatten(), m[1:].flatten()]))[0,1].item())
