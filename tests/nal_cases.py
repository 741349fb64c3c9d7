"""Adversarial payloads for the NAL emulation-prevention pass + a plain-Python statement of the reference's automaton
(/root/reference/src/h264-lab.h:3926-3975 nal_count_esc / nal_put_esc)."""
import numpy as np


def escape_ref(p: bytes) -> bytes:
    out = bytearray(b"\x00\x00\x00\x01")
    cntz = 0
    for b in p:
        if cntz == 2 and b <= 3:
            out.append(3)
            cntz = 0
        cntz = 0 if b else cntz + 1
        out.append(b)
    return bytes(out)


def cases():
    rng = np.random.RandomState(7)
    c = [b"\x65", b"\x65\x00", b"\x65\x00\x00", b"\x65\x00\x00\x00", b"\x65\x00\x00\x01", b"\x65\x00\x00\x03\x00\x00\x03",
         b"\x00" * 1, b"\x00" * 2, b"\x00" * 3, b"\x00" * 7, b"\x00" * 255, b"\x00" * 256, b"\x00" * 257, b"\x00" * 1000,
         b"\x61" + b"\x00\x00\x04" * 100, b"\x61" + b"\x00\x00\x02" * 100, bytes(range(256)) * 3]
    # triples straddling the 256-byte block edges at every alignment
    for edge in (254, 255, 256, 257, 258, 511, 512, 513):
        for pat in (b"\x00\x00\x00", b"\x00\x00\x01", b"\x00\x00\x03", b"\x00\x00\x00\x00\x00", b"\x00\x00\x04"):
            for lead in range(3):
                base = bytearray(rng.randint(4, 256, size=900).astype(np.uint8).tobytes())
                at = edge - lead
                base[at:at + len(pat)] = pat
                c.append(bytes(base))
    # random data rich in zeros
    for n in (5, 63, 64, 65, 300, 1023, 1024, 1025, 5000, 40000):
        for pz in (0.3, 0.6, 0.9):
            a = rng.randint(0, 6, size=n).astype(np.uint8)
            a[rng.rand(n) < pz] = 0
            c.append(a.tobytes())
    return c
