import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle.gp_oracle import OracleGP
X, y = bench.make_c2_data(8192)
print("cpu_count", os.cpu_count())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' ")
def one(n):
    o = OracleGP(X[:n], y[:n])
    o.params[o.ls_key].fill_(-1.0); o.params["covar_module.raw_outputscale"].fill_(0.3)
    o.params["likelihood.noise_covar.raw_noise"].fill_(-6.0); o.params["mean_module.constant"].fill_(0.4)
    t0 = time.perf_counter(); o.loss_and_grad(); return time.perf_counter() - t0
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    one(512)
    for n in (2048, 4096):
        print(th, n, round(one(n), 2), flush=True)
