import torch, sys, os
sys.path.insert(0, os.getcwd())
from early_exit_transformer_amd.frontend import MelFrontend
from oracle import frontend_ref as FR
L=4000
t=torch.arange(L)/16000.0
wave=torch.sin(2*torch.pi*1000.0*t)
want=FR.mel_frontend(wave)
got=MelFrontend()(wave.cuda()).cpu()
print(want.shape, got.shape)
print("want[:,10][:12]", want[:12,10]); print("got [:,10][:12]", got[:12,10])
print("ratio", (got[:,10]/want[:,10])[:12])
# power spectrum check through a delta: impulse at sample
imp=torch.zeros(L); imp[1000]=1.0
w2=FR.mel_frontend(imp); g2=MelFrontend()(imp.cuda()).cpu()
print("impulse frames nonzero want", (w2.sum(0)>0).nonzero().flatten().tolist(), "got", (g2.sum(0)>0).nonzero().flatten().tolist())
print(w2[:5,6], g2[:5,6])
