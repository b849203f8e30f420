import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sqeazy_amd
from sqeazy_amd import synth
from oracle import sqy_oracle as o
vol = synth.stack((64, 256, 256))
rc, blob = sqeazy_amd.encode("lz4", vol, nthreads=2)
want = o.pipeline_encode("lz4", vol)
print(rc, len(blob), len(want))
hs = sqeazy_amd.header_size(want)
a = np.frombuffer(blob, np.uint8); b = np.frombuffer(want, np.uint8)
m = min(a.size, b.size)
d = np.nonzero(a[:m] != b[:m])[0]
d = d[d >= hs]; print("first diff at", d[:5], "header", hs)
# walk frames of the oracle output to locate the chunk
raw = vol.view(np.uint8).reshape(-1)
off = hs; k = 0
while off < len(want):
    size = int.from_bytes(want[off+7:off+11], 'little'); body = size & 0x7fffffff
    end = off + 11 + body + 4
    if d.size and off <= d[0] < end:
        print("chunk", k, "frame at", off, "body", body, "raw?", bool(size >> 31), "diff offset in body", d[0] - off - 11)
        gsz = int.from_bytes(blob[off+7:off+11], 'little')
        print("gpu size field", gsz & 0x7fffffff, bool(gsz >> 31))
        # decode both bodies as far as possible and compare sequences
        def seqs(block):
            i=0; n=len(block); out=[]; pos=0
            while i<n:
                tok=block[i]; i+=1; lit=tok>>4
                if lit==15:
                    while True:
                        s=block[i]; i+=1; lit+=s
                        if s!=255: break
                i+=lit
                if i>=n: out.append((pos,lit,None,None)); break
                offv=block[i]|(block[i+1]<<8); i+=2
                ml=tok&15
                if ml==15:
                    while True:
                        s=block[i]; i+=1; ml+=s
                        if s!=255: break
                out.append((pos,lit,offv,ml+4)); pos+=lit+ml+4
            return out
        sw = seqs(want[off+11:off+11+body]); 
        try:
            sg = seqs(blob[off+11:off+11+(gsz&0x7fffffff)])
        except Exception as e:
            sg = []; print("gpu parse error", e)
        for i,(x,y) in enumerate(zip(sw,sg)):
            if x!=y:
                print("seq", i, "oracle", x, "gpu", y, "prev", sw[i-1] if i else None); break
        break
    off = end; k += 1
