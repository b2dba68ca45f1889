import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    oa, ga, wa, ba = a[k]; ob, gb, wb, bb = b[k]
    rep0 = a[(k[0], k[1], k[2], k[3], 0)]
    same_run = torch.equal(oa, rep0[0]) and torch.equal(ga, rep0[1])
    dg = (ga.float() - gb.float()).abs()
    nz = (dg > 0).float().mean().item()
    rel = dg.max().item() / gb.float().abs().max().item()
    print(k, "out max diff %.2e" % (oa - ob).abs().max().item(), "| ghead: %.4f of elements differ, max diff / max %.2e, rel L2 %.2e" % (nz, rel, (dg.norm() / gb.float().norm()).item()),
          "| gw rel %.2e gb rel %.2e" % (((wa - wb).norm() / wb.norm()).item(), ((ba - bb).abs() / bb.abs()).item()), "| A repeatable:", same_run)
