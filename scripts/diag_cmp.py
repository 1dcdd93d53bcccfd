import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    if a[k].shape != b[k].shape:
        print(k, "shape", a[k].shape, b[k].shape); continue
    d = (a[k].double() - b[k].double()).abs()
    print(k, "max", float(d.max()), "ndiff", int((d > 0).sum()), "of", d.numel(), "nonfinite", int((~torch.isfinite(a[k])).sum()), int((~torch.isfinite(b[k])).sum()))
