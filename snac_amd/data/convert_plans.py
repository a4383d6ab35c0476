"""Convert the reference's joblib plan datasets into one compact integer archive.

Run in the build container only (needs /root/reference + joblib):

    PYTHONDONTWRITEBYTECODE=1 python snac_amd/data/convert_plans.py

Input : Env/{1D,2D,3D}/data_*_envplan_500_{train,val,test}.pkl  (python lists of float64 arrays;
        loaded by the reference at e.g. Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:29-30)
Output: snac_amd/data/plans.npz with keys
          1d_sin_<split>      int16 [P, 30]       plan heights
          2d_<dens>_<split>   uint8 [P, 26, 26]   {0,1}, full bordered grid as stored by the reference
          3d_<dens>_<split>   uint8 [P, 26, 26]   {0,6}
        plus sha1_<key> digests (over the int8 bytes of the full table) that tests pin.
The values are checked to be exact small integers before the cast.
"""
import hashlib
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("SNAC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans.npz")


def main():
    import joblib

    out = {}
    for split in ("train", "val", "test"):
        a = np.asarray(joblib.load(os.path.join(REF, "Env/1D/data_1d_dynamic_sin_envplan_500_%s.pkl" % split)))
        assert a.ndim == 2 and a.shape[1] == 30 and np.all(a == np.round(a)) and a.min() >= 0 and a.max() < 128
        out["1d_sin_%s" % split] = a.astype(np.int16)
        for dim in (2, 3):
            for dens in ("dense", "sparse"):
                p = os.path.join(REF, "Env/%dD/data_%dd_dynamic_%s_envplan_500_%s.pkl" % (dim, dim, dens, split))
                a = np.asarray(joblib.load(p))
                assert a.shape[1:] == (26, 26) and np.all(a == np.round(a))
                vals = set(np.unique(a).tolist())
                assert vals <= ({0.0, 1.0} if dim == 2 else {0.0, 6.0}), vals
                # the 3-wide frame of every stored plan is empty
                inner = np.zeros((26, 26), bool)
                inner[3:23, 3:23] = True
                assert np.all(a[:, ~inner] == 0)
                out["%dd_%s_%s" % (dim, dens, split)] = a.astype(np.uint8)
    digests = {}
    for k, v in out.items():
        digests["sha1_" + k] = np.array(hashlib.sha1(v.astype(np.int8).tobytes()).hexdigest())
    out.update(digests)
    np.savez_compressed(OUT, **out)
    for k in sorted(out):
        if not k.startswith("sha1_"):
            print(k, out[k].shape, out[k].dtype, str(digests["sha1_" + k])[:12])
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
