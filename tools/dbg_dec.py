import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import audio_codec_amd as A
from oracle import pac_oracle as po, pac_oracle_vq as pv
gold = np.load('tests/golden/excerpt_vq_harpsichord.npz')
data = bytes(gold['pac_vq128'])
cp, pos = A.pacfile.parse_header(data)
enc = A.context.encoder_for_params(cp)
offs, sizes = [], []
while pos < len(data):
    n = int.from_bytes(data[pos:pos+4],'little'); offs.append(pos+4); sizes.append(n); pos += 4+n
body = torch.frombuffer(bytearray(data)+bytearray(8), dtype=torch.uint8).to(enc.device)
out = enc.decode_vq(body, torch.tensor(sizes,dtype=torch.int32,device=enc.device), cp.nChannels, offsets=torch.tensor(offs,dtype=torch.int64,device=enc.device), want_lines=True, want_pcm=False)
st = out['status'].cpu().numpy(); print('status', np.unique(st))
lines = out['lines'].cpu().numpy(); fl = out['flags'].cpu().numpy(); ba = out['bit_alloc'].cpu().numpy()
p = po.make_params(cp.sampleRate, cp.nChannels, 128); p.useVQ=True; p.useSBR=False; p.omittedBands=[]
i = 4
br = po.BitReader(data[offs[i]:offs[i]+sizes[i]]+b'\0'*8); print('flags', br.get(3), fl[i]); ov = br.get(4)
alloc = [a+1 if a else 0 for a in (br.get(12) for _ in range(p.sfBands.nBands))]
print('alloc', alloc); print('gpu  ', ba[i][:17].tolist(), 'ov', ov, out['overall'].cpu().numpy()[i][0])
want = pv.decode_lines_vq(br, p, alloc, False, False)
for b in range(p.sfBands.nBands):
    lo, hi = p.sfBands.lowerLine[b], p.sfBands.upperLine[b]+1
    d = np.max(np.abs(lines[i][lo:hi]-want[lo:hi])); m = np.max(np.abs(want[lo:hi]))
    print(b, hi-lo, alloc[b], 'maxdiff %.3e max %.3e' % (d, m), 'norm ratio %.6f' % (np.linalg.norm(lines[i][lo:hi])/max(np.linalg.norm(want[lo:hi]),1e-300)))
