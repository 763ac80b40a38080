"""Bit-reproducibility of the HIP path on a LOADED device.

Round 2 ended red because the two-rank step on one GPU differed from the single-process step in single BatchNorm channels.  The
cause was not a race of ours but a gfx950 quirk: a packed-fp32 VALU instruction whose OP_SEL routes the high dword of a 64-bit
VGPR source to the low result lane (the compiler's own choice for channel 1 of every group of 8 in ``bn_bwd_apply_cb_v4_kernel``)
occasionally read 0.0 for one 16-lane pass when a second process computed on the same CUs.  These tests
keep the condition that exposed it: a neighbour PROCESS running the same kernels, started before this process's children touch
the GPU -- every output must equal the solo run bit for bit."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _tool(name, args, env=None, timeout=900):
    e = dict(os.environ, MCDSEG_PRETRAINED="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", name)] + args, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("shape,dil,iters", [("4,64,48,64", 1, 300),     # 64-channel layers: tr64 weight gradient, 64-row tiles
                                             ("4,16,96,128", 1, 200),    # thin full-resolution layers: LDS-window kernels
                                             ("4,256,24,32", 2, 200),    # 128x128 tiles, tr128 weight gradient
                                             ("16,512,60,80", 4, 40)])   # the benchmark's 256x128 forward / dgrad / tr256 tiles
def test_fused_groups_are_bitwise_next_to_a_computing_neighbour(shape, dil, iters):
    """three fused conv+BN+ReLU groups (residual, block-internal, plain) forward + backward in a loop, in TWO processes at once:
    every output and every BatchNorm-backward intermediate of every iteration equals iteration 0"""
    _need_gpu()
    r = _tool("op_contention.py", ["--procs", "2", "--iters", str(iters), "--shape", shape, "--dil", str(dil)])
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-3000:]
    done = [ln for ln in r.stdout.splitlines() if "done:" in ln]
    assert len(done) == 2 and all("done: 0 of" in ln for ln in done), r.stdout[-6000:]


@pytest.mark.parametrize("op,shape,iters", [("half", "4,256,45,80", 100),     # round 6: a Bottleneck chain in 2-byte storage (16-bit epilogues, bf16 sums)
                                            ("loss", "4,41,30,40", 200),      # the fused up-sampler + loss kernel (hidden LDS-DMA, counted vmcnt)
                                            ("up8_bwd", "4,41,30,40", 200)])  # the up-sampler's backward band kernel (hidden LDS-DMA)
def test_hand_waited_kernels_and_the_two_byte_chain_are_bitwise_under_load(op, shape, iters):
    """``tools/op_contention.py --op``: the kernels whose waits are hand-counted around LDS-DMAs the compiler does not see, and round 6's
    2-byte chain, each in a loop in TWO processes on the one device, every output of every iteration bit for bit that of iteration 0
    (the long form at the benchmark's sizes: tools/run_soak.sh, profiles/r06_determinism_soak.log)"""
    _need_gpu()
    r = _tool("op_contention.py", ["--op", op, "--procs", "2", "--iters", str(iters), "--shape", shape])
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-3000:]
    done = [ln for ln in r.stdout.splitlines() if "done:" in ln]
    assert len(done) == 2 and all("done: 0 of" in ln for ln in done), r.stdout[-6000:]


@pytest.mark.parametrize("mode,reps", [("grads", 3), ("step", 2)])  # (round 3 ran 5 / 3 repetitions: 90 s of the suite; the offending
# instruction has since been banned from the library by a CPU disassembly test, so this is a regression guard, not a soak)
def test_model_step_is_bitwise_with_two_ranks_and_a_copy_loop_on_the_device(mode, reps):
    """drn_d_38 at 4 x 6 x 192 x 256 (round 2's failure showed in 6 of 10 such runs): two ranks on the one device, and two ranks next
    to a process streaming 1 GB device copies -- full tensors (every gradient / every parameter and buffer after the step) against the
    solo run"""
    _need_gpu()
    r = _tool("dp_repro.py", ["--mode", mode, "--reps", str(reps), "--cases", "world2,stress2"], env={"DP_SIZE": "4,192,256"}, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-3000:]
    assert "=== 0 run(s) differ" in r.stdout, r.stdout[-6000:]
