"""GPU, two ranks on ONE device (gloo through host memory): the PRODUCT's multi-rank start-up and sharding on hardware — `AudioToken.load_encoder` distributes the
model (acoustic: a checkpoint file only rank 0 reads -> one flat broadcast; semantic_s: rank 0's finalized model -> packed export -> broadcast -> `packed=` import on
rank 1), every rank passes the start-up probe, `encode_batch_files` LPT-shards ONE shared directory, and the token files are byte-identical to a single-process run
(tools/n8_shared_dir.py asserts all of it; the 8-rank form is tools/n8_rehearsal.sh). RCCL itself needs one device per rank: not available on a one-GPU box."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tokenizer", ["acoustic", "semantic_s"])
def test_product_path_world2_on_one_device(cuda_device, tokenizer):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    with tempfile.TemporaryDirectory(prefix="at_dist_gpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as work:
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                            "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "tools", "n8_shared_dir.py"), work, tokenizer, "14", "2"],
                           env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = p.stdout
    assert "all 2 ranks encode it alike" in out and "byte-identical: True" in out, out[-1500:]
    print(out)
