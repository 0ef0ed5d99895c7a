import os, sys, random
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from pcramp_amd import api, words as W
from oracle_lib import Oracle
import test_gpu_seed_scan as T
o = Oracle()
rng = random.Random(991)
seqs, pairs = T._border_case(rng, o)
thr = float(np.float32(1.0) * np.float32(0.9))
a = T._screener(None); b = T._screener(2)
e3 = T._entries(a, seqs, pairs, thr); e2 = T._entries(b, seqs, pairs, thr)
s3, s2 = set(e3), set(e2)
print("missing in seed2:", [(W.word_text((x[0], x[1])), x[2:], len(seqs[x[3]])) for x in sorted(s2 - s3)])
print("extra in seed2:", [(W.word_text((x[0], x[1])), x[2:], len(seqs[x[3]])) for x in sorted(s3 - s2)])
for pi, p in enumerate(pairs):
    for x in sorted(s2 - s3):
        w = (x[0], x[1])
        for side in (0, 1):
            for ww in (p[side],):
                cnt = o.word_and(ww, w)
                if cnt >= 16: print("pair", pi, side, W.word_text(ww), cnt, o.word_size(ww))
