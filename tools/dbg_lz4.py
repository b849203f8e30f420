import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sqeazy_amd
from sqeazy_amd import synth
from oracle import sqy_oracle as o
pipe = sys.argv[1] if len(sys.argv) > 1 else "lz4"
shape = tuple(int(x) for x in sys.argv[2].split("x")) if len(sys.argv) > 2 else (64, 256, 256)
vol = synth.stack(shape)
rc, blob = sqeazy_amd.encode(pipe, vol, nthreads=2)
want = o.pipeline_encode(pipe, vol)
if pipe != "lz4":      # the bytes LZ4 sees: payload of the same pipeline without the sink
    pre = o.pipeline_encode(pipe.rsplit("->", 1)[0], vol)
    lzin = np.frombuffer(pre, np.uint8)[sqeazy_amd.header_size(pre):]
else:
    lzin = vol.view(np.uint8).reshape(-1)
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
                print("seq", i, "oracle", x, "gpu", y, "prev", sw[i-2:i] if i else None)
                base = k * 262144; q = x[0] + x[1]
                print("bytes at oracle match pos", q, bytes(lzin[base+q-8:base+q+24]).hex())
                if x[2]: print("oracle cand", q - x[2], bytes(lzin[base+q-x[2]-8:base+q-x[2]+24]).hex())
                if y[2]: qg = y[0] + y[1]; print("gpu match pos", qg, "cand", qg - y[2], bytes(lzin[base+qg-y[2]-8:base+qg-y[2]+24]).hex())
                break
        break
    off = end; k += 1
