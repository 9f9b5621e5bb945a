import sys, time, os
sys.path.insert(0, '.')
import torch
from oracle.umgen_oracle import OracleUMGen
from umgen_amd.config import large_config
from umgen_amd.weights import synthetic_state_dict
full = large_config()
one = type(full)(**{**full.__dict__, "n_ego_tar_layer": 1, "n_ego_ca_layer": 1, "n_map_tar_layer": 1, "n_box_tar_layer": 1, "n_tar_layer": 1, "n_oar_layer": 1})
o = OracleUMGen(one, synthetic_state_dict(one, seed=0))
E=768
with torch.no_grad():
    for th in (8, 16, 32, 64, 128):
        torch.set_num_threads(th)
        x = torch.randn(1, 20, 1031, E)
        t0=time.perf_counter(); o._block_tar(x, "transformer.TAR.0"); t1=time.perf_counter()
        kv = (torch.randn(1,1100,E), torch.randn(1,1100,E)); xx = torch.randn(1,1,E)
        o._block_oar(xx, "transformer.OAR.0", kv)
        t2=time.perf_counter()
        for _ in range(8): o._block_oar(xx, "transformer.OAR.0", kv)
        t3=time.perf_counter()
        print(th, "threads: blockTAR(S=1031)", round(t1-t0,2), "s; OAR step", round((t3-t2)/8*1e3,2), "ms", flush=True)
