"""Per-kernel resource usage (VGPRs, spills, scratch, LDS) read from the gfx950 code object inside libfsv2v_hip.so.

The shared library embeds a clang offload bundle (section .hip_fatbin); its device entry is an ELF whose NT_AMDGPU_METADATA note
is a msgpack document with one record per kernel.  No ROCm tool is needed: plain struct + msgpack parsing.

    python tools/kernel_meta.py [path/to/lib.so]      -> one line per kernel
"""
import os
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_elfs(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        cur = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "amdgcn" in triple and size:
                yield triple, blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def _notes(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "64-bit ELF expected"
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type = struct.unpack_from("<I", elf, sh + 4)[0]
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if sh_type != 7:                      # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name.rstrip(b"\0"), ntype, desc


def kernels(path):
    """list of dicts (name, vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size)"""
    blob = open(path, "rb").read()
    out = []
    for triple, elf in _device_elfs(blob):
        for name, ntype, desc in _notes(elf):
            if name == b"AMDGPU" and ntype == 32:
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    out.append({key.lstrip("."): val for key, val in k.items()
                                if key in (".name", ".vgpr_count", ".agpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
                                           ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size")})
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "few-shot-vid2vid_amd", "libfsv2v_hip.so")
    ks = kernels(path)
    for k in sorted(ks, key=lambda k: k["name"]):
        print("%-100s vgpr %3d  spill %d/%d  scratch %4d B  lds %6d B" % (
            k["name"][:100], k["vgpr_count"], k["vgpr_spill_count"], k["sgpr_spill_count"], k["private_segment_fixed_size"],
            k["group_segment_fixed_size"]))
    print("%d kernels" % len(ks))
