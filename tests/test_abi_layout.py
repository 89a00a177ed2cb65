"""The Python view of the C ABI (`pgsd/_lib.py`: hand-written ctypes.Structure classes) is tied to
`include/pgsd.h` field by field.

Every struct the header declares is parsed out of it, a C program generated from that parse prints
sizeof / offsetof / member size as the COMPILER sees them, and each line must agree with the ctypes class of
the same struct: same field names in the same order, same offsets, same sizes.  A field inserted into (or
removed from, or retyped in) the header fails the test until `_lib.py` follows; so does a new struct
without a ctypes twin."""
import ctypes
import os
import re
import subprocess

import product

HEADER = os.path.join(product.ROOT, "include", "pgsd.h")
PRIVATE = os.path.join(product.CSRC, "pgsd_private.h")       # bare-kernel job structs (tests, tools, the binding)

# struct in the header -> ctypes.Structure in pgsd/_lib.py
TWINS = {
    "pgsd_header": "Header", "pgsd_index_entry": "IndexEntry", "pgsd_index_buffer": "IndexBuffer",
    "pgsd_byte_buffer": "ByteBuffer", "pgsd_name_buffer": "NameBuffer", "pgsd_handle": "Handle",
    "pgsd_exchange_stats": "ExchangeStats", "pgsd_comm": "Comm", "pgsd_field_desc": "FieldDesc",
    "pgsd_chunk_req": "ChunkReq", "pgsd_device_config": "DeviceConfig", "pgsd_device_stats": "DeviceStats",
    "pgsd_pack_job": "PackJob", "pgsd_field_dst": "FieldDst", "pgsd_unpack_job": "UnpackJob",
}


def header_structs(text=None):
    """{struct name: [field names in declaration order]} parsed from the header's text."""
    if text is None:
        text = open(HEADER).read() + open(PRIVATE).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"\bstruct\s+(pgsd_\w+)\s*\{(.*?)\}\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fp = re.search(r"\(\s*\*\s*(\w+)\s*\)\s*\(", decl)         # function pointer member
            if fp:
                fields.append(fp.group(1))
                continue
            name = re.search(r"(\w+)\s*(\[[^\]]*\])?\s*$", decl)
            assert name, decl
            fields.append(name.group(1))
        out[m.group(1)] = fields
    return out


def compiler_layout(structs, header_dir, tmp_path):
    """{struct: (sizeof, [(field, offset, size)])} as gcc lays the header's structs out."""
    lines = ['#include "pgsd_private.h"', "#include <stddef.h>", "#include <stdio.h>", "int main(void) {"]
    for sname, fields in structs.items():
        lines.append('printf("S %s %%zu\\n", sizeof(struct %s));' % (sname, sname))
        for f in fields:
            lines.append('printf("F %s %s %%zu %%zu\\n", offsetof(struct %s, %s), sizeof(((struct %s*)0)->%s));'
                         % (sname, f, sname, f, sname, f))
    lines += ["return 0; }"]
    src = os.path.join(str(tmp_path), "layout.c")
    exe = os.path.join(str(tmp_path), "layout")
    with open(src, "w") as fh:
        fh.write("\n".join(lines))
    subprocess.check_call(["gcc", "-std=gnu99", "-I", header_dir, src, "-o", exe])
    out = {}
    for ln in subprocess.check_output([exe], text=True).splitlines():
        p = ln.split()
        if p[0] == "S":
            out[p[1]] = (int(p[2]), [])
        else:
            out[p[1]][1].append((p[2], int(p[3]), int(p[4])))
    return out


def ctypes_layout(cls):
    return ctypes.sizeof(cls), [(n, getattr(cls, n).offset, getattr(cls, n).size) for n, _ in cls._fields_]


def mismatches(layout):
    from pgsd import _lib
    bad = []
    for sname, (size, fields) in layout.items():
        twin = TWINS.get(sname)
        if twin is None:
            bad.append("struct %s has no ctypes twin in pgsd/_lib.py" % sname)
            continue
        csize, cfields = ctypes_layout(getattr(_lib, twin))
        if csize != size:
            bad.append("%s: sizeof %d in C, %d in ctypes" % (sname, size, csize))
        if [f[0] for f in fields] != [f[0] for f in cfields]:
            bad.append("%s: fields %r in C, %r in ctypes" % (sname, [f[0] for f in fields], [f[0] for f in cfields]))
            continue
        for (n, off, sz), (_, coff, csz) in zip(fields, cfields):
            if (off, sz) != (coff, csz):
                bad.append("%s.%s: offset/size %d/%d in C, %d/%d in ctypes" % (sname, n, off, sz, coff, csz))
    return bad


def test_every_struct_of_the_header_matches_its_ctypes_twin(tmp_path):
    structs = header_structs()
    assert set(structs) == set(TWINS), sorted(set(structs) ^ set(TWINS))
    inc = tmp_path / "both"
    inc.mkdir()
    for h in (HEADER, PRIVATE):
        (inc / os.path.basename(h)).write_text(open(h).read())
    layout = compiler_layout(structs, str(inc), tmp_path)
    assert layout["pgsd_header"][0] == 256 and layout["pgsd_index_entry"][0] == 32       # on-disk structs
    assert not mismatches(layout), "\n".join(mismatches(layout))


def test_a_field_inserted_into_the_header_is_noticed(tmp_path):
    """The check has teeth: a header with one more member in pgsd_handle / pgsd_pack_job / a new struct fails."""
    text = open(HEADER).read()
    private = open(PRIVATE).read()
    inc = tmp_path / "inc"
    inc.mkdir()
    for needle, extra, expect in (
            ("        uint64_t cur_frame;\n", "        uint32_t sneaked_in;\n", "pgsd_handle"),
            ("        void* dst;         /* device pointer, N*M elements of dst_type", "        uint64_t sneaked_in;\n", "pgsd_pack_job"),
            ("    struct pgsd_device_stats\n", "    struct pgsd_new_thing { int a; };\n", "pgsd_new_thing")):
        assert needle in text or needle in private
        mutated = text.replace(needle, extra + needle, 1)
        mutated_private = private.replace(needle, extra + needle, 1)
        (inc / "pgsd.h").write_text(mutated)
        (inc / "pgsd_private.h").write_text(mutated_private)
        layout = compiler_layout(header_structs(mutated + mutated_private), str(inc), tmp_path)
        bad = mismatches(layout)
        assert any(expect in b for b in bad), (expect, bad)
