import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_train_parity import _small_problem
from oracle import train as otrain
hw1 = tuple(int(v) for v in os.environ.get("HW1", "48,64").split(","))
hw2 = tuple(int(v) for v in os.environ.get("HW2", "32,24").split(","))
B = int(os.environ.get("B", "48"))
eng, params, x1, x2 = _small_problem("mutopia_ccal_cont", B, hw1, hw2, seed=int(os.environ.get("SEED", "5")))
if os.environ.get("SWAPSCALE"):
    x1 = (x1 * 2).astype(np.float32); x2 = (x2 / 2).astype(np.float32)
loss, corr = eng.train_step(x1, x2, lr=0.002)
p64 = [p.astype(np.float64) for p in params]
o_loss, o_corr, o_grads, o_newp, _ = otrain.loss_and_grads(x1.astype(np.float64), x2.astype(np.float64), p64)
print("loss", loss, o_loss)
for gi, pi in enumerate(otrain.TRAINABLE):
    g = eng.debug_train_tensor("grad", 0, pi).reshape(params[pi].shape) + 2e-5 * params[pi]
    ref = o_grads[gi]
    err = np.abs(g - ref).max() / max(1e-7, np.abs(ref).max())
    print("param %2d tower %d block %d kind %s shape %-16s relerr %.2e  |ref|max %.2e" % (
        pi, pi // 45 + 1, (pi % 45) // 5 + 1, "W b g".split()[pi % 5], str(params[pi].shape), err, np.abs(ref).max()))
# dH check
H1, st1, c1, ls1 = otrain.tower_forward_train(x1.astype(np.float64), p64[0:45])
