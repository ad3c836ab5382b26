"""Golden vectors from the REFERENCE's own host RNG: Caitlyn/Rnd.h compiled from where it lies into
oracle/_ref/librndref.so (oracle/ref_rnd.cpp, `make -C oracle ref`).  Run in the BUILD container only:

    python tests/golden/make_ref_rnd_fixture.py

Writes tests/golden/ref_rnd.json: the first 256 randf2() values from the header's initial state (s_RndState = 1, Rnd.h:7) as
float bit patterns with the state after each call — frame k's `randomVector` is values 2k-2 and 2k-1 (Scene.h:1208) — a run from
a few other states, and PCG_Hash of a set of inputs.  DATA produced by running the reference's code, not its text.
"""
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rndref      # noqa: E402


def bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def main():
    assert rndref.available(), "oracle/_ref/librndref.so is missing: run `make -C oracle ref` (needs /root/reference)"
    L = rndref.lib()
    out = {"source": "Caitlyn/Rnd.h compiled in place (oracle/ref_rnd.cpp)", "sequences": {}, "pcg_hash": {}}
    for start in (1, 0, 2, 0xdeadbeef, 0xffffffff, 123456789):
        L.ref_rnd_set_state(start)
        seq = []
        for _ in range(256 if start == 1 else 16):
            v = L.ref_randf2()
            seq.append([bits(v), L.ref_rnd_state()])
        out["sequences"][str(start)] = seq
    for x in list(range(0, 20)) + [0x1234, 0x1234 ^ 77, 0x7fffffff, 0x80000000, 0xffffffff, 747796405, 2891336453, 277803737, 0xa8beea3c, 0xe92a518a]:
        out["pcg_hash"][str(x)] = L.ref_pcg_hash(x)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_rnd.json"), "w"))
    print("frame 1 randomVector:", struct.unpack("<f", struct.pack("<I", out["sequences"]["1"][0][0]))[0],
          struct.unpack("<f", struct.pack("<I", out["sequences"]["1"][1][0]))[0], "states", hex(out["sequences"]["1"][0][1]), hex(out["sequences"]["1"][1][1]))


if __name__ == "__main__":
    main()
