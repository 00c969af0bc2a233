import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
from style_transfer_amd import weights as W
from style_transfer_amd.attack import AttackRunner, AttackConfig
from style_transfer_amd.data import AttackSet
GOLD='/root/repo/tests/golden'
g=np.load(GOLD+'/attack_notnll.npz')
sd=W.expand_ae_state_dict(W.load_state_dict_file(GOLD+'/surrogate_ae_yelp.safetensors'))
dev=torch.device('cuda:0')
for fmt in ('f16x3','f32'):
    ae=W.AEWeights(sd, dev, gemm_format=fmt)
    r=AttackRunner(ae, None, AttackConfig(), dev)
    sents=[[int(t) for t in row if t!=0] for row in g['enc_src']]
    lat=r.encode(sents).cpu().numpy()
    ref=g['enc_latent']
    rel=np.abs(lat-ref).max(1)/np.abs(ref).max(1)
    print(fmt, 'max rel err vs reference latents %.2e  median %.2e'%(rel.max(), np.median(rel)))
