"""CPU: the oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer: oracle/selftest.c
calls every entry point once, including the degenerate inputs the parity tests feed it."""
import os
import subprocess

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_oracle_selftest_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", ORACLE, "selftest"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    out = subprocess.run([os.path.join(ORACLE, "selftest")], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr[-4000:]
    assert "oracle selftest ok" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
