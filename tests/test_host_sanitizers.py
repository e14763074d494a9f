"""The multi-threaded HOST preparation of the library (counting-sort transposes, validated CSR
copies, the parallel libstdc++ random stream, the kNN per-row pass - irspack_amd/csrc/
host_prep.hpp, knn_host_prep.hpp) compiled on its own with g++ and run under ThreadSanitizer and
AddressSanitizer + UBSan ("race detection").  GPU sanitizers are not
available on the pool; this covers the code that runs on host threads.  The harness
(tests/san/host_prep_san.cpp) also compares every multi-threaded result with a sequential one,
and the parallel random stream with std::normal_distribution<float> itself (> 2^18 values)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "san", "host_prep_san.cpp")


def _build_and_run(tmp_path, flags, env_extra):
    exe = str(tmp_path / "host_prep_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", *flags, SRC, "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, **env_extra)
    return subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_preparation_under_thread_sanitizer(tmp_path):
    r = _build_and_run(tmp_path, ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"})
    if "unexpected memory mapping" in r.stderr or "personality" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container: " + r.stderr[:200])
    assert r.returncode == 0, r.stderr[-4000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert "host_prep_san ok" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_preparation_under_address_and_ub_sanitizers(tmp_path):
    r = _build_and_run(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                  "-fno-omit-frame-pointer"],
                       {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "host_prep_san ok" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_preparation_plain_optimised_build(tmp_path):
    """The same harness without a sanitizer at -O3: this is the build in which the bulk Mersenne
    Twister runs its AVX2 clones (the sanitizer builds take the plain loops - an ifunc resolver
    runs before a sanitizer's runtime is up), compared word for word with std::mt19937 and
    variate for variate with std::normal_distribution."""
    r = _build_and_run(tmp_path, ["-O3"], {})
    assert r.returncode == 0, r.stderr[-4000:]
    assert "host_prep_san ok" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("san", ["thread", "address"])
def test_cpu_oracle_under_sanitizers(san):
    """`make -C oracle SAN=thread|address san`: the multi-threaded CPU oracle (atomic-cursor row
    dispatch, per-thread Gramian partials; two epochs of each solver on 4 threads)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), f"SAN={san}", "san"])
    exe = os.path.join(ROOT, "oracle", "_san", f"oracle_san_{san}")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert r.returncode == 0, r.stderr[-4000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "oracle_san ok" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("portable", [False, True])
def test_mt19937_jump_ahead_arithmetic(tmp_path, portable):
    """Products modulo MT19937's characteristic polynomial against a bit-by-bit reference, the tabulated
    polynomial against Berlekamp-Massey, jumps against the engine's own sequence - with the carry-less
    multiplier and with the portable product a host without PCLMULQDQ takes (mt_jump.hpp)."""
    exe = str(tmp_path / "mt_jump_check")
    src = os.path.join(ROOT, "tests", "san", "mt_jump_check.cpp")
    flags = ["-DIRS_MTJUMP_NO_CLMUL"] if portable else []
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", *flags, src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mt_jump_check ok" in r.stdout
    if portable:
        assert "portable product" in r.stdout
