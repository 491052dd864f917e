"""CPU sanitizer builds: the host-only C++ of the product (BlockManager, Scheduler incl. chunked prefill, tokenizer,
config validation) and the C oracle, compiled with -fsanitize=address,undefined and RUN here.  CPU only — never on the GPU box's
device code (GPU AddressSanitizer is not available there)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_host_code_under_asan_ubsan(seed):
    csrc = os.path.join(ROOT, "nano-vllm-rs_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(csrc, "build", "host_selftest_asan"), str(seed)], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "host_selftest ok" in r.stdout, r.stderr[-2000:]


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    libubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    env = dict(ENV, LD_PRELOAD=f"{libasan}:{libubsan}", NVO_ORACLE_LIB=os.path.join(ROOT, "oracle", "libnvr_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", OMP_NUM_THREADS="2")       # (the interpreter itself leaks at exit)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_oracle_smoke.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0 and "asan oracle smoke ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
