"""The gfx950 code objects inside libirspack_amd.so, read without a GPU: per-kernel scratch, registers
and LDS from the AMDGPU metadata notes.

Why: `#pragma unroll` gives up silently above LLVM's -pragma-unroll-threshold.  In round 4 the block loop
of the K = 128 Cholesky solve crossed it after an unrelated change: the loop stayed a loop, the 36
accumulator tiles moved to scratch memory behind a run-time index (7,854 scratch instructions, 816 bytes
per lane), and every parity test still passed.  Scratch is the visible symptom, so it is pinned here.
"""
import os
import struct

import msgpack
import pytest

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "irspack_amd",
                   "libirspack_amd.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(elf):
    """(name, type, offset, size) of the sections of a 64-bit little-endian ELF image"""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2 and elf[5] == 1
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    raw = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + i * shentsize)
        raw.append((name, typ, off, size))
    stroff = raw[shstrndx][2]
    out = []
    for name, typ, off, size in raw:
        end = elf.index(b"\0", stroff + name)
        out.append((elf[stroff + name:end].decode(), typ, off, size))
    return out


def _device_images(lib_bytes):
    fat = next((off, size) for name, _t, off, size in _sections(lib_bytes) if name == ".hip_fatbin")
    blob = lib_bytes[fat[0]:fat[0] + fat[1]]
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                yield triple, blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + 1)


def _kernels(image):
    for name, typ, off, size in _sections(image):
        if typ != 7:  # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", image, p)
            p += 12
            owner = image[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = image[p:p + descsz]
            p += (descsz + 3) & ~3
            if owner == b"AMDGPU" and ntype == 32:  # NT_AMDGPU_METADATA
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    yield k


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("libirspack_amd.so is not built")
    data = open(LIB, "rb").read()
    ks = {}
    for triple, image in _device_images(data):
        assert "gfx950" in triple, triple  # one architecture only
        for k in _kernels(image):
            ks[k[".name"]] = k
    assert len(ks) > 100
    return ks


def test_solve_kernels_keep_their_accumulators_in_registers(kernels):
    solve = {n: k for n, k in kernels.items() if "ials_solve_kernel" in n}
    assert len(solve) >= 20
    worst = max(solve.values(), key=lambda k: k[".private_segment_fixed_size"])
    # a handful of spilled registers at most (56 bytes at the time of writing); the accumulators of the
    # K = 128 kernels alone would be 576
    assert worst[".private_segment_fixed_size"] <= 128, (worst[".name"], worst[".private_segment_fixed_size"])


def test_no_kernel_spills_by_the_kilobyte(kernels):
    worst = max(kernels.values(), key=lambda k: k[".private_segment_fixed_size"])
    assert worst[".private_segment_fixed_size"] <= 1024, (worst[".name"], worst[".private_segment_fixed_size"])


def test_headline_kernel_fits_four_waves_per_simd(kernels):
    # unit-confidence Cholesky, K = 64: <= 128 registers and no scratch (scratch costs ~60 us per launch),
    # 16 waves x 9.9 KB of LDS per compute unit
    k = next(v for n, v in kernels.items()
             if "ials_solve_kernel" in n and "ILi4ELi0ELi0ELb1ELb0ELb0E" in n)
    assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 128, k[".vgpr_count"]
    assert k[".private_segment_fixed_size"] == 0
    assert 16 * k[".group_segment_fixed_size"] <= 160 * 1024


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _vregs(token):
    import re
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", token):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain is not installed")
def test_no_dpp_read_within_two_wait_states_of_a_vector_write(tmp_path):
    """gfx9: a vector instruction that writes a register and a DPP instruction that reads it through its
    lane-permuting operand need two wait states between them, and the hardware does not interlock.  The
    compiler keeps that for its own DPP instructions; the row operations of the Cholesky diagonal tile
    (ials_chol16.hpp) are inline assembly, which the hazard recogniser does not look into - their
    `s_nop 1` covers what the source order puts in front of them, this test covers what the register
    allocator might (a copy placed right before the assembly).  Fall-through order, every kernel."""
    import re
    import subprocess

    if not os.path.exists(LIB):
        pytest.skip("libirspack_amd.so is not built")
    data = open(LIB, "rb").read()
    n_dpp, bad = 0, []
    for i, (_triple, image) in enumerate(_device_images(data)):
        elf = tmp_path / f"image{i}.elf"
        elf.write_bytes(image)
        text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(elf)], capture_output=True,
                              text=True, check=True).stdout
        hist = []  # (registers written by a vector instruction, wait states the instruction is worth)
        for line in text.split("\n"):
            m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*(//.*)?$", line)
            if not m or line.rstrip().endswith(":"):
                continue
            op, parts = m.group(1), [p.strip() for p in m.group(2).split(",")]
            if "_dpp" in op and len(parts) >= 2:
                n_dpp += 1
                src = _vregs(parts[1].split(" ")[0])
                ws = 0
                for written, cost, txt in reversed(hist):
                    if ws >= 2:
                        break
                    if any(r in written for r in src):
                        bad.append((txt, line.strip()))
                        break
                    ws += cost
            if op == "s_nop":
                hist.append((frozenset(), int(parts[0], 0) + 1, line.strip()))
            elif op.startswith("v_") and not op.startswith("v_cmp"):
                hist.append((frozenset(_vregs(parts[0])), 1, line.strip()))
            else:
                hist.append((frozenset(), 1, line.strip()))
            del hist[:-4]
    assert n_dpp > 1000  # (the scan found the instructions it is about)
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain is not installed")
def test_no_use_of_a_permute_result_before_its_wait(tmp_path):
    """The gather loop of the solve kernels issues `ds_bpermute_b32` from inline assembly with an immediate
    offset and places the `s_waitcnt lgkmcnt(0)` itself, one sub-step later (ials_kernels.hpp: perm_issue /
    fetch_finish): the compiler does not know the result is in flight in between.  Should the register
    allocator ever put a copy or a spill of that register there, the gathered index would be read before
    the LDS unit delivered it - silently wrong rows, in one build and not in another.  This scan of every
    kernel's disassembly (fall-through order) checks that no instruction touches the destination register of
    a `ds_bpermute_b32` before the next wait that drains the LDS queue."""
    import re
    import subprocess

    if not os.path.exists(LIB):
        pytest.skip("libirspack_amd.so is not built")
    data = open(LIB, "rb").read()
    n_perm, bad = 0, []
    for i, (_triple, image) in enumerate(_device_images(data)):
        elf = tmp_path / f"image{i}.elf"
        elf.write_bytes(image)
        text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(elf)], capture_output=True,
                              text=True, check=True).stdout
        queue = []  # outstanding LDS / scalar-memory operations in issue order: (permute's register or None, line)
        for line in text.split("\n"):
            m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*(//.*)?$", line)
            if not m or line.rstrip().endswith(":"):
                queue.clear()  # a label: another basic block / kernel
                continue
            op, rest = m.group(1), m.group(2)
            if op == "s_waitcnt":
                # lgkmcnt(N): all but the N youngest operations of the queue have returned (LDS returns in order)
                w = re.search(r"lgkmcnt\((\d+)\)", rest)
                if w:
                    del queue[:max(0, len(queue) - int(w.group(1)))]
                elif rest.strip() in ("0", "0x0"):
                    queue.clear()
                continue
            if op == "s_endpgm":
                queue.clear()
                continue
            pending = {reg: where for reg, where in queue if reg is not None}
            if pending:
                used = set(_vregs(rest))
                for reg, where in pending.items():
                    if reg in used:
                        bad.append((where, line.strip()))
                        queue[:] = [(r, wl) for r, wl in queue if r != reg]
            if op == "ds_bpermute_b32":
                n_perm += 1
                dst = _vregs(rest.split(",")[0])
                queue.append((dst[0] if dst else None, line.strip()))
            elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                queue.append((None, line.strip()))
    assert n_perm > 500  # (the scan found the instructions it is about)
    assert not bad, bad[:5]
