import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from early_exit_transformer_amd import synth
from early_exit_transformer_amd.model import Early_conformer, exit_ctc_losses
from oracle.conformer_ref import xavier_like_reference
for passes in (3, 1):
    m = Early_conformer(device="cuda", **bench.CFG); xavier_like_reference(m); m = m.cuda().train(); m.train_passes = passes
    opt = torch.optim.AdamW(m.parameters(), lr=5e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.1)
    mel = synth.synth_mel(32, 80, 1027, seed=1).cuda(); lens = torch.full((32,), 1027)
    tgt, tl = synth.synth_targets(32, 42, 256, seed=1); tgt, tl = tgt.cuda(), tl.cuda()
    torch.manual_seed(0); ls = []
    for i in range(40):
        opt.zero_grad(set_to_none=True)
        loss = exit_ctc_losses(m(mel, lens), tgt, tl).sum(); loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0); opt.step(); ls.append(loss.item())
    print(f"passes {passes}: loss", " ".join(f"{v:.1f}" for v in ls[::4]), "| last grad norm", float(gn), "| finite", all(map(lambda v: v == v, ls)))
    m.eval()
    with torch.no_grad(): out = m(mel[:4], lens[:4])
    print("   eval after training finite:", torch.isfinite(out).all().item(), "max logp", out.max().item())
