import numpy as np, sys
a = np.fromfile(sys.argv[1], np.float32).reshape(-1, 768)
b = np.fromfile(sys.argv[2], np.float32).reshape(-1, 768)
n = min(len(a), len(b))
d = np.abs(a[:n] - b[:n]).max(axis=1)
bad = np.argwhere(d > 0).ravel()
print("steps", len(a), len(b), "differing steps:", len(bad), bad[:10].tolist(), "max", d.max())
if len(bad):
    j = bad[0]
    w = np.argwhere(a[j] != b[j]).ravel()
    print("first diff step", j, "n elems", len(w), w[:20].tolist(), a[j][w[:5]], b[j][w[:5]])
