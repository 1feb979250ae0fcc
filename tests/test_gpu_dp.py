"""-m gpu: the data-parallel plumbing on one MI355X -- torch tensors aliasing the library's device
buffers (__cuda_array_interface__), an RCCL (backend "nccl") all-reduce over them with world_size 1,
and training continuing correctly afterwards."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402
import pytest  # noqa: E402

import oracle as O  # noqa: E402
from common import assert_close, random_csr
from gpu_common import gpu_fm, to_gpu

pytestmark = pytest.mark.gpu


def test_alias_and_rccl_allreduce_world1():
    """Runs in a fresh interpreter: torch must initialise its HIP runtime before libnimfm_hip.so is
    loaded (as in bench.py); the other GPU tests of this process loaded the library first."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "dp world1 ok" in out.stdout


def _main():
    import torch
    import torch.distributed as dist

    import nimfm_amd as nf
    from nimfm_amd import dp

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, d, m, k, B = 4000, 300, 8, 16, 256
        Xo = random_csr(n, d, m, seed=4)
        rng = np.random.default_rng(2)
        y = rng.standard_normal(n)
        P0, w0 = rng.standard_normal((1, k, d)) * 0.05, np.zeros(d)
        X = to_gpu(Xo)
        X.set_targets(y)
        for solver in ("sgd", "adagrad"):
            fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
            opt = (nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B) if solver == "sgd"
                   else nf.newAdaGrad(maxIter=1, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B))
            opt._handle(fm, X.ctx, "minibatch")
            views = dp.ParamViews(torch, dev, fm, opt)
            # the alias really is the library's memory: P in the device layout [d][Kp], Kp = 16
            fm._pull()
            assert_close(views.params[0].cpu().numpy().reshape(d, k).T, fm.P[0], 0, 0)
            for _ in range(3):
                opt._epoch(X, None, 0, n)
                opt.it += n
                views.average(dist, 1, force=True)  # one replica: the exchange must be the identity
            opt._finalize_into(fm)
            if solver == "sgd":
                P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
                for _ in range(3):
                    b, it, _, _ = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, O.sgd_cfg(), B, it=it)
            else:
                cfg = O.adagrad_cfg()
                P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
                st = O.AdaState(1, d, k, d)
                for _ in range(3):
                    b, it, _, _ = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, B, st, it=it)
                b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
            assert abs(fm.intercept - b) < 1e-11
            assert_close(fm.w, w, 1e-9, 1e-12, solver + " w")
            assert_close(fm.P, P, 1e-9, 1e-12, solver + " P")
    finally:
        dist.destroy_process_group()
    print("dp world1 ok")


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    _main()
