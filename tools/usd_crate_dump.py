#!/usr/bin/env python3
"""Dev-only reader for binary USD ("crate", PXR-USDC 0.8.0) files -- just enough to read the physics parameters of the
reference's rover asset without ``pxr`` (absent from the build image):

    python tools/usd_crate_dump.py /root/reference/rover_envs/assets/robots/aau_rover_simple/rover_instance.usd
    python tools/usd_crate_dump.py --model-fixture     # check tools/derive_rover_model.py's tables against the asset

Format notes: SURVEY.md Appendix E (header + TOC, TfFastCompression = LZ4 block(s), delta-coded integer arrays,
TOKENS / STRINGS / FIELDS / FIELDSETS / PATHS / SPECS sections, 64-bit value representations).  Standard library only.
"""
from __future__ import annotations

import json
import os
import struct
import sys


# ------------------------------------------------------------------------------------------------- LZ4 block
def lz4_block(src: bytes, max_out: int | None = None) -> bytes:
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        tok = src[i]
        i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = src[i]
                i += 1
                lit += b
                if b != 255:
                    break
        out += src[i:i + lit]
        i += lit
        if i >= n:
            break
        off = src[i] | (src[i + 1] << 8)
        i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = src[i]
                i += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        start = len(out) - off
        for k in range(ml):
            out.append(out[start + k])
    return bytes(out)


def fast_decompress(buf: bytes) -> bytes:
    """TfFastCompression: first byte = chunk count; 0 => the rest is ONE raw LZ4 block."""
    nchunks = buf[0]
    if nchunks == 0:
        return lz4_block(buf[1:])
    out = bytearray()
    p = 1
    for _ in range(nchunks):
        (sz,) = struct.unpack_from("<i", buf, p)
        p += 4
        out += lz4_block(buf[p:p + sz])
        p += sz
    return bytes(out)


def decode_ints(raw: bytes, n: int) -> list:
    """USD integer coding: int32 common delta, 2-bit codes (0 common, 1 int8, 2 int16, 3 int32), running sums."""
    (common,) = struct.unpack_from("<i", raw, 0)
    ncode = (n * 2 + 7) // 8
    codes = raw[4:4 + ncode]
    p = 4 + ncode
    out, acc = [], 0
    for i in range(n):
        c = (codes[i // 4] >> ((i % 4) * 2)) & 3
        if c == 0:
            d = common
        elif c == 1:
            (d,) = struct.unpack_from("<b", raw, p)
            p += 1
        elif c == 2:
            (d,) = struct.unpack_from("<h", raw, p)
            p += 2
        else:
            (d,) = struct.unpack_from("<i", raw, p)
            p += 4
        acc = (acc + d) & 0xFFFFFFFF
        out.append(acc - (1 << 32) if acc & 0x80000000 else acc)
    return out


class Reader:
    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.p = data, pos

    def u64(self):
        (v,) = struct.unpack_from("<Q", self.d, self.p)
        self.p += 8
        return v

    def take(self, n):
        b = self.d[self.p:self.p + n]
        self.p += n
        return b

    def compressed_ints(self, n):
        size = self.u64()
        return decode_ints(fast_decompress(self.take(size)), n)


# ------------------------------------------------------------------------------------------------- crate file
class Crate:
    def __init__(self, path: str):
        self.data = d = open(path, "rb").read()
        assert d[:8] == b"PXR-USDC", "not a USD crate file"
        self.version = tuple(d[8:11])
        (toc,) = struct.unpack_from("<q", d, 16)
        (nsec,) = struct.unpack_from("<Q", d, toc)
        self.sections = {}
        for i in range(nsec):
            off = toc + 8 + i * 32
            name = d[off:off + 16].split(b"\0")[0].decode()
            start, size = struct.unpack_from("<qq", d, off + 16)
            self.sections[name] = (start, size)
        self._tokens()
        self._fields()
        self._fieldsets()
        self._paths()
        self._specs()

    def _tokens(self):
        r = Reader(self.data, self.sections["TOKENS"][0])
        count, raw_size, comp_size = r.u64(), r.u64(), r.u64()
        raw = fast_decompress(r.take(comp_size))
        self.tokens = [t.decode("utf-8", "replace") for t in raw.split(b"\0")][:count]
        r = Reader(self.data, self.sections["STRINGS"][0])
        n = r.u64()
        self.strings = list(struct.unpack_from(f"<{n}I", self.data, r.p))

    def _fields(self):
        r = Reader(self.data, self.sections["FIELDS"][0])
        n = r.u64()
        tok = r.compressed_ints(n)
        size = r.u64()
        reps = fast_decompress(r.take(size))
        self.fields = [(self.tokens[tok[i]], struct.unpack_from("<Q", reps, 8 * i)[0]) for i in range(n)]

    def _fieldsets(self):
        r = Reader(self.data, self.sections["FIELDSETS"][0])
        n = r.u64()
        self.fieldsets = r.compressed_ints(n)

    def _paths(self):
        r = Reader(self.data, self.sections["PATHS"][0])
        npaths, nenc = r.u64(), r.u64()
        pidx, tidx, jump = r.compressed_ints(nenc), r.compressed_ints(nenc), r.compressed_ints(nenc)
        self.paths = [None] * npaths

        def build(i, parent):
            while True:
                tok = tidx[i]
                is_prop = tok < 0
                name = self.tokens[abs(tok)]
                if parent is None:
                    path = "/"
                elif is_prop:
                    path = parent + "." + name
                else:
                    path = (parent if parent.endswith("/") else parent + "/") + name
                self.paths[pidx[i]] = path
                j = jump[i]
                has_child = j > 0 or j == -1
                has_sibling = j >= 0
                if has_child:
                    if has_sibling:
                        build(i + j, parent)          # sibling subtree
                    parent, i = path, i + 1             # descend into the child (next entry)
                    continue
                if has_sibling:                       # sibling only: it is simply the next entry of the stream
                    i = i + 1
                    continue
                return

        sys.setrecursionlimit(10000)
        build(0, None)

    def _specs(self):
        r = Reader(self.data, self.sections["SPECS"][0])
        n = r.u64()
        p, f, t = r.compressed_ints(n), r.compressed_ints(n), r.compressed_ints(n)
        self.specs = {}
        for i in range(n):
            fields = {}
            k = f[i]
            while self.fieldsets[k] != -1:
                name, rep = self.fields[self.fieldsets[k]]
                fields[name] = rep
                k += 1
            self.specs[self.paths[p[i]]] = (t[i], fields)

    # ---------------------------------------------------------------------------------------------- values
    def value(self, rep: int):
        is_array, inlined, compressed = bool(rep >> 63 & 1), bool(rep >> 62 & 1), bool(rep >> 61 & 1)
        ty = (rep >> 48) & 0xFF
        payload = rep & ((1 << 48) - 1)
        d = self.data
        if is_array:
            if payload == 0:
                return []
            (n,) = struct.unpack_from("<Q", d, payload)
            base = payload + 8
            fmt = {3: "i", 8: "f", 9: "d", 5: "q"}.get(ty)
            if fmt and not compressed:
                return list(struct.unpack_from(f"<{n}{fmt}", d, base))
            if ty in (23, 24) and not compressed:
                f, w = ("d", 8) if ty == 23 else ("f", 4)
                flat = struct.unpack_from(f"<{3 * n}{f}", d, base)
                return [flat[3 * i:3 * i + 3] for i in range(n)]
            if ty == 11 and not compressed:
                return [self.tokens[i] for i in struct.unpack_from(f"<{n}I", d, base)]
            return f"<array type {ty} x{n}{' compressed' if compressed else ''}>"
        if ty == 1:
            return bool(payload & 1)
        if ty == 3:
            return struct.unpack("<i", struct.pack("<I", payload & 0xFFFFFFFF))[0]
        if ty == 8 or (ty == 9 and inlined):
            return struct.unpack("<f", struct.pack("<I", payload & 0xFFFFFFFF))[0]
        if ty == 9:
            return struct.unpack_from("<d", d, payload)[0]
        if ty == 10:
            return self.tokens[self.strings[payload]]
        if ty in (11, 12):
            return self.tokens[payload]
        if ty in (16, 17):          # quatd / quatf, stored x, y, z, w -> return (w, x, y, z)
            if inlined:
                return "<inlined quat>"
            f = "d" if ty == 16 else "f"
            x, y, z, w = struct.unpack_from(f"<4{f}", d, payload)
            return (w, x, y, z)
        if ty in (23, 24, 25):      # vec3d / vec3f / vec3h
            if inlined:
                return tuple(float(v) for v in struct.unpack("<3b", struct.pack("<I", payload & 0xFFFFFF)[:3]))
            f = {23: "d", 24: "f", 25: "e"}[ty]
            return tuple(struct.unpack_from(f"<3{f}", d, payload))
        if ty == 42:
            return {0: "def", 1: "over", 2: "class"}.get(payload, payload)
        if ty == 44:
            return {0: "varying", 1: "uniform"}.get(payload, payload)
        return f"<type {ty}{' inlined' if inlined else ''} payload {payload}>"

    def attr(self, prim: str, name: str, default=None):
        spec = self.specs.get(f"{prim}.{name}")
        if spec is None or "default" not in spec[1]:
            return default
        return self.value(spec[1]["default"])

    def prims(self):
        return sorted(p for p in self.specs if "." not in p.rsplit("/", 1)[-1])


# ------------------------------------------------------------------------------------------------- rover model
ROVER_USD = "/root/reference/rover_envs/assets/robots/aau_rover_simple/rover_instance.usd"


def rover_tables(path=ROVER_USD):
    c = Crate(path)
    joints, links = {}, {}
    for p in c.prims():
        name = p.rsplit("/", 1)[-1]
        if c.attr(p, "physics:localPos0") is not None and c.attr(p, "physics:localRot0") is not None:
            joints[name] = {
                "path": p,
                "localPos0": c.attr(p, "physics:localPos0"), "localPos1": c.attr(p, "physics:localPos1"),
                "localRot0": c.attr(p, "physics:localRot0"), "localRot1": c.attr(p, "physics:localRot1"),
                "axis": c.attr(p, "physics:axis"),
                "lowerLimit": c.attr(p, "physics:lowerLimit"), "upperLimit": c.attr(p, "physics:upperLimit"),
                "stiffness": c.attr(p, "drive:angular:physics:stiffness"), "damping": c.attr(p, "drive:angular:physics:damping"),
                "maxForce": c.attr(p, "drive:angular:physics:maxForce"),
            }
        mass = c.attr(p, "physics:mass")
        if mass is not None:
            links[name] = {"path": p, "mass": mass, "centerOfMass": c.attr(p, "physics:centerOfMass")}
    return c, joints, links


def check_model_fixture():
    """Compare the tables transcribed in tools/derive_rover_model.py with what the asset actually contains."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import derive_rover_model as drm
    c, joints, links = rover_tables()
    worst = 0.0
    for jname, (parent, p0, p1, r0, r1) in drm.JOINTS.items():
        cand = [k for k in joints if k.startswith(jname)]
        assert cand, f"joint {jname} not found in the asset ({sorted(joints)})"
        j = joints[cand[0]]
        for ours, theirs in ((p0, j["localPos0"]), (p1, j["localPos1"])):
            worst = max(worst, max(abs(a - b) for a, b in zip(ours, theirs)))
        for ours, theirs in ((r0, j["localRot0"]), (r1, j["localRot1"])):
            # quaternions are sign-ambiguous
            e = min(max(abs(a - b) for a, b in zip(ours, theirs)), max(abs(a + b) for a, b in zip(ours, theirs)))
            worst = max(worst, e)
        assert j["axis"] == "X", (jname, j["axis"])
    total = 0.0
    for lname, (mass, com, _) in drm.LINKS.items():
        assert lname in links, f"link {lname} not found in the asset"
        assert abs(links[lname]["mass"] - mass) < 1e-6, (lname, links[lname]["mass"], mass)
        total += links[lname]["mass"]
        tc = links[lname]["centerOfMass"]
        if tc is not None and not isinstance(tc, str):
            worst = max(worst, max(abs(a - b) for a, b in zip(com, tc)))
    print(f"{len(drm.JOINTS)} joints, {len(drm.LINKS)} links checked against {ROVER_USD}")
    print(f"total mass {total} kg; worst |transcribed - asset| over joint frames / COMs = {worst:.2e}")
    return worst


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--model-fixture":
        worst = check_model_fixture()
        sys.exit(0 if worst < 2e-4 else 1)
    path = sys.argv[1] if len(sys.argv) > 1 else ROVER_USD
    c = Crate(path)
    print(f"{path}: crate {c.version}, {len(c.paths)} paths, {len(c.fields)} fields, {len(c.tokens)} tokens")
    if "--json" in sys.argv:
        _, joints, links = rover_tables(path)
        print(json.dumps({"joints": joints, "links": links}, indent=1, default=str))
        return
    for p in sorted(c.specs):
        t, fields = c.specs[p]
        if "default" in fields:
            print(f"{p} = {c.value(fields['default'])}")


if __name__ == "__main__":
    main()
