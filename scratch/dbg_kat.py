import numpy as np, sys
sys.path.insert(0, ".")
from imageclust_amd import _lib
ctx = _lib.Context(0)
E = np.array([[0], [1], [3], [7], [8], [20]], np.float32)
try:
    print(ctx.cluster(E, 1, 2))
except Exception as e:
    print("ERR", e)
