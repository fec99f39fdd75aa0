"""CPU-side sanitizer runs (there is no GPU sanitizer on this pool): the oracle's own test file executed against
an AddressSanitizer + UndefinedBehaviorSanitizer build of oracle/mip_oracle.c (`make -C oracle asan`). The glTF
extractor's sanitizer run is tests/test_gltf_fuzz.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, MIP_ORACLE_LIBRARY=os.path.join(ROOT, "oracle", "_build", "libmip_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), os.path.join(ROOT, "tests", "test_fuzz.py"),
                        "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"], capture_output=True, text=True, errors="replace", timeout=900, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert " passed" in r.stdout
