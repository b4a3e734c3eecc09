import sys, numpy as np, json
a, b = sys.argv[1], sys.argv[2]
fa, fb = json.load(open(a)), json.load(open(b))
first = next((i for i, (x, y) in enumerate(zip(fa, fb)) if x != y), None)
print("first differing step:", first, "of", len(fa))
for t in range(int(sys.argv[3])):
    A, B = np.load(a + ".step%d.npz" % t), np.load(b + ".step%d.npz" % t)
    for k in A.files:
        x, y = A[k], B[k]
        if x.tobytes() != y.tobytes():
            d = np.argwhere(x != y)
            print("step", t, k, "differs at", len(d), "places; first", d[:5].tolist(), "values", [(x[tuple(i)], y[tuple(i)]) for i in d[:5]])
