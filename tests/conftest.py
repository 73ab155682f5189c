import os
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


# a SIGABRT raised anywhere in the test process (runtime, allocator, library) leaves its C backtrace here
(ROOT / "gpurun_out").mkdir(exist_ok=True)  # (open(O_CREAT) in a signal handler creates files, not directories)
os.environ.setdefault("SA_HIP_ABORT_TRACE", str(ROOT / "gpurun_out" / "abort_backtrace.txt"))
# ... and a GPU memory fault (ROCr: "Memory access fault by GPU node-N", then abort) leaves the GPU core dump next to it instead
# of in the working directory of a box that is gone afterwards: `rocgdb -c` names the kernel and the faulting wave
# (the second abort on record had its message but not its core)
os.environ.setdefault("HSA_COREDUMP_PATTERN", str(ROOT / "gpurun_out" / "gpucore.%p"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference checkout at /root/reference (skipped elsewhere)")


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def sa():
    import sequencealigner_amd
    sequencealigner_amd.load_library()
    return sequencealigner_amd
