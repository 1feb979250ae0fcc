"""bench.py --gpus N outside torchrun starts N ranks itself: the parent process makes no GPU call
and imports neither torch nor the library; the ranks are torch.distributed.run children.  On a box without a GPU every
rank stops at "needs an MI355X".  torchrun ends the surviving ranks as soon as the first one fails, so the test does not
wait for every rank to reach that message: each rank announces itself before its imports, and the test counts the distinct
announcements (and that at least one rank got as far as the refusal)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_gpus_flag_spawns_that_many_ranks():
    import torch

    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher (on a GPU box the ranks would really run)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=560, env=env)
    txt = out.stdout + out.stderr
    assert out.returncode != 0
    import re

    ranks = set(re.findall(r"\[bench\.py\] rank (\d+) of 2 started", txt))
    assert ranks == {"0", "1"}, txt[-3000:]
    assert txt.count("bench.py needs an MI355X") >= 1, txt[-3000:]


def test_parent_does_not_touch_the_gpu_before_spawning():
    """static check: spawn_ranks runs before `import torch` / `import nimfm_amd` in main()"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("return spawn_ranks(args)") < main.index("import torch")
    body = src[src.index("def spawn_ranks(args):"):src.index("def make_dataset(")]
    code = [ln.strip() for ln in body.splitlines()]
    assert not any(ln.startswith(("import torch", "from torch", "import nimfm_amd", "from nimfm_amd")) for ln in code)
