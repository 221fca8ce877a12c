"""The conv kernels retire their global -> LDS loads with COUNTED vmcnt waits (igemm8_kernel, wgrad8_kernel, the 3-stage
igemm_kernel): a register spill is a scratch access that counts in vmcnt too, so a spilling build would miscount its
loads without failing to compile.  This reads the gfx950 code objects out of the built libyfs.so (no GPU, no
recompilation) and checks the metadata of every conv kernel: no scratch, no vector-register spills, at most 256 VGPRs."""
import os
import struct

import pytest

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yolo_from_scratch_amd", "libyfs.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """Every gfx950 ELF inside the clang offload bundles of a host object / shared library."""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernel_metadata(elf):
    """amdhsa.kernels of one code object (NT_AMDGPU_METADATA note, msgpack)."""
    import msgpack
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        stype, = struct.unpack_from("<I", elf, sh + 4)
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if stype != 7:   # SHT_NOTE
            continue
        q = off
        while q < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, q)
            name = elf[q + 12:q + 12 + namesz]
            d = q + 12 + ((namesz + 3) & ~3)
            if ntype == 32 and name.startswith(b"AMDGPU"):
                return msgpack.unpackb(elf[d:d + descsz], raw=False).get("amdhsa.kernels", [])
            q = d + ((descsz + 3) & ~3)
    return []


def test_conv_kernels_use_no_scratch():
    if not os.path.exists(LIB):
        pytest.skip("libyfs.so not built")
    blob = open(LIB, "rb").read()
    seen = 0
    for elf in code_objects(blob):
        for k in kernel_metadata(elf):
            name = k[".name"]
            if not any(t in name for t in ("igemm8_kernel", "wgrad8_kernel", "igemm_kernel", "wgrad_kernel", "stem_pool_kernel")):
                continue
            seen += 1
            assert k[".private_segment_fixed_size"] == 0, f"{name}: {k['.private_segment_fixed_size']} B of scratch per lane"
            assert k.get(".vgpr_spill_count", 0) == 0, f"{name} spills vector registers"   # (SGPRs spill to VGPR lanes: no memory)
            assert k[".vgpr_count"] <= 256, name
    assert seen >= 40, f"only {seen} conv kernels found in {LIB}: the bundle layout may have changed"


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not found")
def test_8phase_k_loops_hold_only_their_own_vmcnt_waits():
    """igemm8_kernel's K-loop keeps two K-steps of global -> LDS loads in flight and waits for them with its own counted
    vmcnt(6) (vmcnt(0) on the last steps).  The compiler adds waits of its own whenever it believes a vector-memory access
    that wrote a register the loop reuses may still be outstanding (it did when the tile loop's exit path merged with its
    back edge: vmcnt(0) at the top of every K-step, 5 % slower, no wrong result), and it wraps a DMA whose scalar offset
    it believes divergent in a waterfall loop (40 % slower).  Both are silent, so the disassembly is checked: between the
    first and the last MFMA of every instantiation, nothing but vmcnt(6) and at most three vmcnt(0), and no waterfall."""
    import subprocess
    import tempfile
    if not os.path.exists(LIB):
        pytest.skip("libyfs.so not built")
    blob = open(LIB, "rb").read()
    checked = 0
    for elf in code_objects(blob):
        if not any("igemm8_kernel" in k[".name"] for k in kernel_metadata(elf)):
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        body, name = {}, None
        for line in asm.splitlines():
            if line.endswith(">:"):
                name = line.split("<", 1)[1][:-2]
                body[name] = []
            elif name is not None:
                body[name].append(line.split("//")[0].strip())
        for name, lines in body.items():
            if "igemm8_kernel" not in name:
                continue
            mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
            assert mf, name
            loop = lines[mf[0]:mf[-1] + 1]
            waits = [l for l in loop if l.startswith("s_waitcnt") and "vmcnt" in l]
            other = [l for l in waits if "vmcnt(6)" not in l and "vmcnt(0)" not in l]
            assert not other, f"{name}: compiler-inserted waits in the K-loop: {sorted(set(other))}"
            assert sum("vmcnt(0)" in l for l in waits) <= 3 and sum("vmcnt(6)" in l for l in waits) == 3, f"{name}: {waits}"
            assert not any(l.startswith("v_readfirstlane") and "s_xor_b64 exec" in " ".join(loop[i:i + 8])
                           for i, l in enumerate(loop)), f"{name}: waterfall loop around an LDS-DMA"
            checked += 1
    assert checked >= 10, f"only {checked} igemm8_kernel instantiations found"
