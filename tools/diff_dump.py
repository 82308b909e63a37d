import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = ~np.isclose(a, b, atol=1e-5, rtol=1e-4, equal_nan=False).all(1)
print("rays", a.shape, "bad", int(bad.sum()), "nan rows", int(np.isnan(b).any(1).sum()))
idx = np.nonzero(bad)[0]
print("first bad", idx[:40]); print("last bad", idx[-10:])
if len(idx): print("example", a[idx[0]][:8], b[idx[0]][:8])
