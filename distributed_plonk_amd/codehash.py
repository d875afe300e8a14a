"""Hashes of the DEVICE CODE of individual kernels, read from the objects hipcc produced (no external tool).

Why: numbers that come from separate profiling runs (PMC counters: HBM traffic, VALU instruction counts — profiles/pmc_current.json)
may only be quoted by bench.py for the kernels they were collected from.  A hash over all kernel sources (build.source_hash) answers that
conservatively, but it is invalidated by every edit anywhere — an error-path fix in the C ABI, a new kernel beside the measured one.  The
machine code of the measured kernel is the thing that must not have changed: an object's `.hip_fatbin` section holds a clang offload bundle,
its gfx950 entry is an ELF code object, and a kernel is a FUNC symbol of that ELF — hash its bytes.
"""
import hashlib
import json
import os
import re
import struct

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(data):
    """-> {name: (offset, size, addr)}, [(name, value, size, info, shndx)] symbols of an ELF64 little-endian image."""
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        raise ValueError("not an ELF64 little-endian image")
    e_shoff, = struct.unpack_from("<Q", data, 0x28)
    e_shentsize, e_shnum, e_shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    if e_shnum == 0 or e_shstrndx == 0xFFFF:             # extended numbering lives in section 0
        sh0 = struct.unpack_from("<IIQQQQIIQQ", data, e_shoff)
        e_shnum = e_shnum or sh0[5]
        e_shstrndx = sh0[6] if e_shstrndx == 0xFFFF else e_shstrndx
    hdrs = [struct.unpack_from("<IIQQQQIIQQ", data, e_shoff + i * e_shentsize) for i in range(e_shnum)]
    str_off = hdrs[e_shstrndx][4]

    def cstr(base, off):
        end = data.index(b"\0", base + off)
        return data[base + off:end].decode()

    secs, symtab = {}, None
    for h in hdrs:
        name = cstr(str_off, h[0])
        secs[name] = (h[4], h[5], h[3])
        if h[1] == 2:                                    # SHT_SYMTAB
            symtab = h
    syms = []
    if symtab is not None:
        strtab = hdrs[symtab[6]][4]
        for i in range(symtab[5] // 24):
            st_name, st_info, _other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", data, symtab[4] + 24 * i)
            syms.append((cstr(strtab, st_name), st_value, st_size, st_info, st_shndx))
    return secs, syms, hdrs


def device_code_object(obj_path, arch="gfx950"):
    """the gfx950 code object (bytes) bundled into a host object compiled by hipcc"""
    with open(obj_path, "rb") as f:
        data = f.read()
    secs, _, _ = _elf_sections(data)
    if ".hip_fatbin" not in secs:
        return None
    off, size, _ = secs[".hip_fatbin"]
    fat = data[off:off + size]
    if fat[:len(BUNDLE_MAGIC)] != BUNDLE_MAGIC:
        raise ValueError(f"{obj_path}: .hip_fatbin is not an uncompressed clang offload bundle")
    n, = struct.unpack_from("<Q", fat, len(BUNDLE_MAGIC))
    at = len(BUNDLE_MAGIC) + 8
    for _ in range(n):
        e_off, e_size, t_len = struct.unpack_from("<QQQ", fat, at)
        triple = fat[at + 24:at + 24 + t_len].decode()
        at += 24 + t_len
        if triple.startswith("hip") and triple.endswith(arch):
            return fat[e_off:e_off + e_size]
    return None


def kernel_symbol_ranges(code_object):
    """{mangled kernel name: (file offset, size)} for every defined FUNC symbol of a device code object"""
    _, syms, hdrs = _elf_sections(code_object)
    out = {}
    for name, value, size, info, shndx in syms:
        if (info & 0xF) != 2 or size == 0 or shndx == 0 or shndx >= len(hdrs):      # STT_FUNC, defined
            continue
        h = hdrs[shndx]
        out[name] = (h[4] + (value - h[3]), size)
    return out


def kernel_symbols(code_object):
    """{mangled kernel name: code bytes}"""
    return {name: code_object[start:start + size] for name, (start, size) in kernel_symbol_ranges(code_object).items()}


def base_name(mangled):
    """_Z15ntt_pass_kernelILi8E... -> ntt_pass_kernel; _ZL20ntt_gen_plane_kernel... -> ntt_gen_plane_kernel; plain C names unchanged"""
    m = re.match(r"^_ZL?(\d+)", mangled)
    if not m:
        return mangled
    n = int(m.group(1))
    return mangled[m.end():m.end() + n]


def kernel_code_hashes(obj_dir, arch="gfx950"):
    """{kernel base name: sha256[:16] over (mangled name, code bytes) of all its instantiations, sorted} over every object in obj_dir"""
    groups = {}
    for f in sorted(os.listdir(obj_dir)):
        if not f.endswith(".o"):
            continue
        co = device_code_object(os.path.join(obj_dir, f), arch)
        if co is None:
            continue
        for name, code in kernel_symbols(co).items():
            groups.setdefault(base_name(name), []).append((name, code))
    out = {}
    for base, items in groups.items():
        h = hashlib.sha256()
        for name, code in sorted(items):
            h.update(name.encode())
            h.update(struct.pack("<Q", len(code)))
            h.update(code)
        out[base] = h.hexdigest()[:16]
    return out


def host_code_hashes(obj_dir):
    """{"host:<unit>": sha256[:16] of the HOST machine code (.text* sections) of each object} and {"unit_of:<kernel>": unit}.  A kernel's
    counters depend on its launch shape as well as on its code — window widths, grids, pass plans are decided by the host code of the unit
    that launches it and by the option defaults compiled into plonk_api.o — so bench.py quotes a kernel from a profile of other
    sources only if these agree too."""
    out = {}
    for f in sorted(os.listdir(obj_dir)):
        if not f.endswith(".o"):
            continue
        path = os.path.join(obj_dir, f)
        with open(path, "rb") as fh:
            data = fh.read()
        secs, _, _ = _elf_sections(data)
        h = hashlib.sha256()
        for name in sorted(secs):
            if name == ".text" or name.startswith(".text."):
                off, size, _ = secs[name]
                h.update(name.encode())
                h.update(data[off:off + size])
        unit = f[:-2]
        out["host:" + unit] = h.hexdigest()[:16]
        co = device_code_object(path)
        if co is not None:
            for name in kernel_symbols(co):
                out["unit_of:" + base_name(name)] = unit
    return out


def write_hashes(obj_dir, out_path):
    hashes = kernel_code_hashes(obj_dir)
    hashes.update(host_code_hashes(obj_dir))
    with open(out_path, "w") as f:
        json.dump(hashes, f, indent=0, sort_keys=True)
    return hashes


if __name__ == "__main__":
    import sys
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "obj")
    for k, v in sorted(kernel_code_hashes(d).items()):
        print(v, k)
