#!/usr/bin/env python
"""tests/golden/e2e_small.npz -> e2e_small.bin and e2e_indoor.npz -> e2e_indoor.bin: flat little-endian containers a C program can
read without zip / npy parsing (tests/c/e2e_small.c, tests/c/e2e_indoor.c).  Pure re-encoding of the golden vectors (inputs + the
reference's outputs); the JSON test_cfg / head_kw strings are expanded into scalar entries.

  file   := magic "IVXF0001" | int32 n_entries | entry*
  entry  := int32 name_len | name bytes | int32 dtype (0 f32, 1 i64, 2 u8) | int32 ndim | int64 shape[ndim] | data
"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def convert(name):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), allow_pickle=False)
    ent = []
    for k in g.files:
        a = g[k]
        if a.dtype.kind == 'U':
            if k.endswith('test_cfg') or k.endswith('head_kw'):
                for ck, cv in json.loads(str(a)).items():
                    if isinstance(cv, (int, float, bool)):
                        ent.append((k + '::' + ck, np.asarray([float(cv)], np.float32)))
            continue
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        ent.append((k, np.ascontiguousarray(a)))
    out = os.path.join(ROOT, 'tests', 'golden', name + '.bin')
    with open(out, 'wb') as f:
        f.write(b'IVXF0001')
        f.write(struct.pack('<i', len(ent)))
        for k, a in ent:
            code = {np.dtype(np.float32): 0, np.dtype(np.int64): 1, np.dtype(np.uint8): 2}[a.dtype]
            kb = k.encode()
            f.write(struct.pack('<i', len(kb)) + kb + struct.pack('<ii', code, a.ndim))
            f.write(struct.pack('<%dq' % a.ndim, *a.shape))
            f.write(a.tobytes())
    print(out, os.path.getsize(out), 'bytes,', len(ent), 'entries')


def main():
    for name in (sys.argv[1:] or ['e2e_small', 'e2e_indoor']):
        convert(name)


if __name__ == '__main__':
    sys.exit(main())
