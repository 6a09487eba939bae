"""TEST INFRASTRUCTURE: makes variants of a reference .bin model by rewriting its configuration dump -- how the tests get an
ignore-case lexer, an ignore-case dictionary or a moore-multi-dfa [wbd] without the reference's compile toolchain.  The automata and
maps of the model are kept byte for byte; only dump 0 (the section -> parameter list multi-map) and, when the model carries one, the
validation dump (sizes + CRC32 of the other dumps) are written anew.

Formats (read from the reference, written here from scratch):
  LDB container   cl/src/FALDB.cpp:24-64      int32 count, int32 offsets[count], dumps
  configuration   cl/src/FAMultiMap_pack.cpp:22-53 + cl/inc/FAChains_pack_triv.h:81-163
                  uint32 max_key, uint32 size_of_offset, offsets[max_key + 1] (big-endian, 0 = no entry, else chain offset + 1),
                  padding to 4, chains = {int32 size_of_value, int32 max_count, then per chain: count, values}
  validation      cl/src/FALDB.cpp:67-116     [global] verify-ldb-bin: the last dump is {0, total size of the other dumps, their CRC32}
"""
import struct
import zlib

FUNC_POS_DICT, FUNC_WBD, FUNC_GLOBAL = 12, 19, 20       # FAFsmConst.h FUNC_* (the values bf_model.h and the oracle use)
PARAM_IGNORE_CASE, PARAM_FSM_TYPE, PARAM_VERIFY_LDB_BIN = 22, 26, 70
TYPE_MOORE_MULTI_DFA = 4
# parameters that take no value: cl/src/FALDB.cpp:119-132, and use-byte-encoding / no-dummy-prefix of [pos-dict] (FADictConfKeeper.cpp)
BOOLEAN_PARAMS = {10, 18, 22, 31, 35, 37, 40, 46, 70, 73, 74}


def read_ldb(path):
    b = open(path, "rb").read()
    count = struct.unpack_from("<i", b, 0)[0]
    offs = list(struct.unpack_from("<%di" % count, b, 4)) + [len(b)]
    return [b[offs[i]:offs[i + 1]] for i in range(count)]


def decode_conf(d):
    max_key, soo = struct.unpack_from("<II", d, 0)
    off = 8 + soo * (1 + max_key)
    off += (-off) % 4
    sov = struct.unpack_from("<i", d, off)[0]
    fmt = {1: "b", 2: "<h", 4: "<i"}[sov]
    conf = {}
    for key in range(max_key + 1):
        vo = int.from_bytes(d[8 + soo * key: 8 + soo * (key + 1)], "big")
        if vo == 0:
            continue
        at = off + vo - 1
        n = struct.unpack_from(fmt, d, at)[0]
        conf[key] = [struct.unpack_from(fmt, d, at + sov * (1 + i))[0] for i in range(n)]
    return conf


def encode_conf(conf):
    max_key = max(conf)
    chains = bytearray(struct.pack("<ii", 4, max(len(v) for v in conf.values())))
    offsets = []
    for key in range(max_key + 1):
        if key not in conf:
            offsets.append(0)
            continue
        offsets.append(len(chains) + 1)
        chains += struct.pack("<i%di" % len(conf[key]), len(conf[key]), *conf[key])
    head = bytearray(struct.pack("<II", max_key, 4))
    for o in offsets:
        head += o.to_bytes(4, "big")
    return bytes(head + chains)


def write_ldb(path, dumps, conf):
    dumps = list(dumps)
    dumps[0] = encode_conf(conf)
    dumps = [d + b"\0" * ((-len(d)) % 4) for d in dumps]
    if PARAM_VERIFY_LDB_BIN in conf.get(FUNC_GLOBAL, []):
        body = b"".join(dumps[:-1])
        dumps[-1] = struct.pack("<III", 0, len(body) & 0xFFFFFFFF, zlib.crc32(body) & 0xFFFFFFFF)
    count = len(dumps)
    at = 4 + 4 * count
    offs = []
    for d in dumps:
        offs.append(at)
        at += len(d)
    with open(path, "wb") as f:
        f.write(struct.pack("<i%di" % count, count, *offs))
        for d in dumps:
            f.write(d)
    return path


def params(vals):
    """[(param, value or None)] of a section's flat list"""
    out, i = [], 0
    while i < len(vals):
        if vals[i] in BOOLEAN_PARAMS:
            out.append((vals[i], None))
            i += 1
        else:
            out.append((vals[i], vals[i + 1]))
            i += 2
    return out


def make_variant(src, dst, section, add_boolean=None, set_param=None):
    """copy of `src` whose `section` also carries the boolean parameter `add_boolean` and / or has `set_param` = (param, value)
    (placed FIRST: fsm-type must precede fsm, FADictConfKeeper / FAWbdConfKeeper read in order)"""
    dumps = read_ldb(src)
    conf = decode_conf(dumps[0])
    vals = list(conf[section])
    if set_param is not None:
        kept = []
        for p, v in params(vals):
            if p != set_param[0]:
                kept += [p] if v is None else [p, v]
        vals = [set_param[0], set_param[1]] + kept
    if add_boolean is not None and add_boolean not in [p for p, _ in params(vals)]:
        vals = [add_boolean] + vals
    conf[section] = vals
    return write_ldb(dst, dumps, conf)
