"""Byte-level walker of dump.bin V3 (reference assembly/hnsw.ts:1080-1124) for tests that craft dumps."""
import struct

HEADER = 48  # magic, version, DIM, M, M_MAX0, EF_CONSTRUCTION, MAX_LAYERS, max_elements, present, entry, max_level, results_cap


def header(dump):
    return list(struct.unpack_from("<12i", dump, 0))


def walk(dump):
    """-> [(id, level, vec_off, [(layer, count_off, ids_off, cap), ...]), ...] in file order."""
    h = header(dump)
    dim, m, m0, present = h[2], h[3], h[4], h[8]
    off, out = HEADER, []
    for _ in range(present):
        nid, lvl = struct.unpack_from("<2i", dump, off)
        vec_off = off + 8
        off = vec_off + dim
        lists = []
        for l in range(lvl + 1):
            cap = m0 if l == 0 else m
            lists.append((l, off, off + 4, cap))
            off += 4 + 4 * cap
        out.append((nid, lvl, vec_off, lists))
    assert off == len(dump), "walker and file length disagree"
    return out


def get_i32(buf, off):
    return struct.unpack_from("<i", buf, off)[0]


def put_i32(buf, off, v):
    struct.pack_into("<i", buf, off, int(v))
