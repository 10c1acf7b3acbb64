import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from infinitevl_amd import _lib
_lib.load(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import kernel_bench
kernel_bench.main()
