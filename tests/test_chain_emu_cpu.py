"""csrc/ba_chain.hpp (the chain solver of the reduced camera system) on the CPU: the kernels compiled with -DCHAIN_HOST_EMU run
on a fiber emulation of a workgroup (tests/host/chain_emu_test.cpp) and are compared with a dense Cholesky solve of the same damped
band + border system, for leaf counts / launch splits / waves per leaf / band widths / with and without intrinsics.  Pins the
solver's index arithmetic; what depends on the hardware's execution is pinned by tests/test_ba_gpu.py."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_chain_solver_emulation_agrees_with_dense_solve():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host"), "chain_emu_test"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HERE, "host", "chain_emu_test"), "quick"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "all shapes agree" in out.stdout
    assert out.stdout.count("ok  ") >= 12, out.stdout
