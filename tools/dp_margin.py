import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_data_parallel as T
B, world, hw1, hw2 = 48, 2, (48, 64), (32, 24)
for rep in range(4):
    params, x1, x2 = T._problem(B, hw1, hw2)
    ref = T._engine(params, hw1, hw2); ref.train_begin(B)
    [ref.train_step(x1, x2, lr=0.002) for _ in range(2)]
    ref_params = ref.get_params(); ref.close()
    EX = T.HostExchange(world); n = B // world
    def rank_body(r):
        eng = T._engine(params, hw1, hw2)
        ar, ag = EX.bind(r, eng)
        eng.comm_init_custom(r, world, ar, ag)
        eng.train_begin(n)
        sl = slice(r * n, (r + 1) * n)
        [eng.train_step(x1[sl], x2[sl], lr=0.002) for _ in range(2)]
        p = eng.get_params(); eng.train_end(); eng.comm_destroy(); eng.close()
        return p
    res = T._run_ranks(world, rank_body)
    worst = (0, -1)
    for i in range(90):
        tol = 1e-4 * max(1.0, float(np.abs(ref_params[i]).max()))
        d = float(np.abs(res[0][i] - ref_params[i]).max()) / tol
        if d > worst[0]: worst = (d, i)
    print("rep %d: worst diff/tol %.3f at param %d" % (rep, worst[0], worst[1]))
