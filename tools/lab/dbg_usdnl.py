"""Lab: the small-loss training loop with every device result checked against stock torch ops, batch by batch."""
import sys, os, importlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.nn.functional as F
from rlvi_amd import driver, ops, _lib
L = _lib.load()
dev = torch.device("cuda:0")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for run in range(runs):
    torch.manual_seed(0)
    xa, ya_noisy, ya_clean, _ = driver.synthetic_digits(5120, noise_rate=0.4, seed=3)
    x, y_noisy = xa[:4096], ya_noisy[:4096]
    xt, yt = xa[4096:], ya_clean[4096:]
    rate = np.ones(12) * 0.4; rate[:4] = np.linspace(0, 0.4, 4)
    m1 = driver.LeNet().to(dev); o1 = torch.optim.SGD(m1.parameters(), lr=float(os.environ.get("LAB_LR", "0.1")), momentum=0.9)
    loader = driver.IndexedLoader(x, y_noisy, 256, shuffle=True, seed=1)
    trace = []
    for epoch in range(12):
        m1.train()
        for bi, (data, labels, _) in enumerate(loader):
            data, labels = data.to(dev), labels.to(dev).long()
            logits = m1(data)
            B = logits.shape[0]
            k = int((1 - rate[epoch]) * B)
            with torch.no_grad():
                lp = ops.per_sample_ce(logits.detach(), labels)
                mask = ops.select_smallest(lp, k)
            loss = ops.selected_cross_entropy(logits, labels, mask, 1.0 / k)
            o1.zero_grad(); loss.backward()
            # checks
            zz = logits.detach().clone().requires_grad_(True)
            ce = F.cross_entropy(zz, labels, reduction='none')
            order = torch.argsort(ce.detach(), stable=True)[:k]
            mref = torch.zeros(B, device=dev); mref[order] = 1
            lref = (ce * mask).sum() / k
            lref.backward()
            gl = torch.autograd.grad(ops.selected_cross_entropy(logits, labels, mask, 1.0 / k), logits, retain_graph=True)[0]
            e_ce = float((lp - ce.detach()).abs().max())
            e_mask = int((mask != mref).sum())
            e_loss = abs(float(loss) - float(lref))
            e_grad = float((gl - zz.grad).abs().max())
            gn = float(torch.cat([p.grad.flatten() for p in m1.parameters()]).norm())
            trace.append((epoch, bi, float(loss), gn))
            if e_ce > 1e-4 or int(mask.sum()) != k or e_loss > 1e-4 * max(1, abs(float(lref))) or e_grad > 1e-5 or not np.isfinite(float(loss)):
                print(f"run {run} epoch {epoch} batch {bi}: ce err {e_ce:.3g} mask diff {e_mask} sum {int(mask.sum())}/{k} loss {float(loss):.6g} ref {float(lref):.6g} grad err {e_grad:.3g}", flush=True)
            o1.step()
    ta = driver.evaluate(driver.IndexedLoader(xt, yt, 512, shuffle=False), m1, dev)
    m1.eval()
    with torch.no_grad():
        tref = 100.0 * float((m1(xt.to(dev)).argmax(1) == yt.to(dev)).sum()) / yt.numel()
    print(f"run {run}: test {ta:.2f} torch {tref:.2f}", flush=True)
    if ta < 85:
        bad += 1
        print("   last 24 batches (epoch, batch, loss, |grad|):", [(e, b, round(l, 4), round(g, 3)) for e, b, l, g in trace[-24:]])
        big = sorted(trace, key=lambda t: -t[3])[:6]
        print("   largest grad norms:", [(e, b, round(l, 4), round(g, 3)) for e, b, l, g in big])
print("collapsed runs:", bad, "of", runs)
