"""GOP sharding of ONE stream over several clip encoders / GPUs.

The reference encodes a stream strictly in order; in constant-QP mode exactly two things cross a key frame: the parity of
idr_pic_id (computable) and enc->mv_clusters, a raster-serial state that every inter macroblock reads (as two rounded start
candidates) and updates (h264-lab.h:5263-5278).  A shard therefore starts from a SPECULATED state, keeps what every
macroblock consumed, and is settled once the shard in front of it is final: H264E_clip_revalidate walks the kept records from the
exact state; the GOPs from the first divergent macroblock on are encoded again.  With row-band slices (slices > 1) the reference
throws the state away after every band (h264-lab.h:6526), so shards are exact as they stand.
"""
from .binding import ClipEncoder


def shard_ranges(nframes, gop, nshards):
    """contiguous GOP-aligned blocks [(first, end)], as even as the GOP count allows"""
    gop = gop or nframes
    ngop = (nframes + gop - 1) // gop
    out, g0 = [], 0
    for k in range(nshards):
        g1 = g0 + (ngop - g0) // (nshards - k)
        if g1 > g0:
            out.append((g0 * gop, min(g1 * gop, nframes)))
        g0 = g1
    return out


class StreamShard:
    """frames [first, end) of a stream on one device; `speculated` = the mv_clusters state assumed in front of `first`"""

    def __init__(self, width, height, first, end, gop, qp, speculated=(0, 0), **kw):
        assert gop and first % gop == 0
        self.first, self.end, self.gop = first, end, gop
        self.enc = ClipEncoder(width, height, end - first, gop=gop, qp=qp, clusters_in=speculated, idr_state=(first // gop) & 1,
                               keep_records=1, **kw)
        self.frames = []            # coded bytes per frame
        self.reencoded = 0

    def first_pass(self):
        out, sizes, st = self.enc.encode()
        self._take(0, out, sizes)
        return st

    def _take(self, frm, out, sizes):
        del self.frames[frm:]
        pos = 0
        for s in sizes:
            self.frames.append(out[pos:pos + s])
            pos += s

    def settle(self, exact_in):
        """make the shard exact for the true state in front of it; returns the exact state behind it"""
        while True:
            frm, state, end_state = self.enc.revalidate(exact_in)
            if frm < 0:
                return end_state
            self.enc.restart(frm, state)
            out, sizes, _ = self.enc.encode(rewind=False)
            self._take(frm, out, sizes)
            self.reencoded += len(sizes)

    def bytes(self):
        return b"".join(self.frames)

    def close(self):
        self.enc.close()
