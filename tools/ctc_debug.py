import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from early_exit_transformer_amd import capi, synth
E,B,T,V,S = 6,5,256,256,42
torch.manual_seed(E*100+T)
logp = torch.log_softmax(torch.randn(E,B,T,V)*2, -1)
tgt, tl = synth.synth_targets(B, S, V, seed=T); tgt = tgt.clone(); tgt[0,2] = tgt[0,1]
lib = capi.load()
lg = logp.cuda(); tg = tgt.cuda(); tlg = tl.cuda()
nll = torch.empty(E*B, device="cuda"); out = torch.empty(E, device="cuda")
lib.eec_ctc_loss(lg.data_ptr(), tg.data_ptr(), tlg.data_ptr(), E,B,T,V,S,0, nll.data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize()
ctc = torch.nn.CTCLoss(blank=0, reduction="none", zero_infinity=False)
il = torch.full((B,), T, dtype=torch.long)
ref = torch.stack([ctc(logp[e].double().permute(1,0,2), tgt, il, tl) for e in range(E)]).reshape(-1)
d = (nll.cpu().double() - ref)
print("target lens", tl.tolist())
for i in range(E*B): print(i//B, i%B, f"gpu {nll[i].item():.5f} ref {ref[i].item():.5f} diff {d[i].item():+.2e}")
