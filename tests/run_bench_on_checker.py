"""TEST-ONLY launcher target (tests/test_parallel.py::test_bench_py_under_the_drivers_launch_line_with_eight_ranks): what the driver runs on the 8-GPU node is
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`; this file is put in bench.py's
place on that command line so that the very same launcher, environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) and argument parsing drive bench.main() on
the host-side kernel checker with the gloo backend - the GPU-less build box cannot run the RCCL path.  Never used by the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

torch.set_num_threads(1)
import conftest  # noqa: E402

conftest.emu_library()
import bench  # noqa: E402

bench.main(sys.argv[1:], checker_device="cpu")
