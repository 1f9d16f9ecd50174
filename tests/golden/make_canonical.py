#!/usr/bin/env python3
"""Adds 'canonical_sha256' (sign bit of zero-magnitude mantissas cleared, see
tests/pac_parse.py) to fullfile.json.  The whole-file bytes are re-created with
the oracle and must first hash to the sha256 the REFERENCE produced
(make_golden.py --full), so the canonical hash is a function of the
reference's own output."""
import hashlib, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pac_oracle as po
from pac_parse import canonical_sha256
from multiprocessing import Pool

def one(key):
    name, tag = key.split(":")
    d = np.load(os.path.join(HERE, f"full_{name}.npz"))
    sr, decl = int(d["sr"]), int(d["declared"])
    data = po.encode_stream(d["pcm"], sr, 128, block_switching=(tag == "bs"), header_samples=decl)
    p = po.make_params(sr, 2, 128)
    c, cleared = canonical_sha256(data, len(po.pac_header(p, decl)), p.sfBands.nLines.tolist(),
                                  p.sfBandsShort.nLines.tolist())
    return key, hashlib.sha256(data).hexdigest(), c, cleared

if __name__ == "__main__":
    path = os.path.join(HERE, "fullfile.json")
    full = json.load(open(path))
    with Pool(8) as pool:
        for key, sha, c, cleared in pool.map(one, sorted(full)):
            assert sha == full[key]["sha256"], key
            full[key]["canonical_sha256"] = c
            full[key]["negative_zero_mantissas"] = cleared
            print(key, c, cleared)
    json.dump(full, open(path, "w"), indent=1, sort_keys=True)
