#!/usr/bin/env python
"""Times gce_fasta_load on an hg19-sized synthetic FASTA (24 contigs of hg19's lengths x --scale, 60-column lines): the literal one-pass
walk (threads = 1) next to the parallel loader.  tools/fasta_bench.py [--scale 1.0] [--threads 0] [--out profiles/x.json]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gencore_amd import capi, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--path", default="/tmp/gce_fasta_bench.fa")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lens = [int(w * a.scale) for w in synth.CONFIGS["cfg3"]["contigs"]]
    if not os.path.exists(a.path) or os.path.getsize(a.path) < sum(lens):
        rng = np.random.default_rng(7)
        with open(a.path, "wb") as f:
            for k, ln in enumerate(lens):
                f.write(b">chr%d synthetic\n" % (k + 1))
                done = 0
                while done < ln:
                    m = min(ln - done, 60 * (1 << 20))
                    bases = np.frombuffer(b"ACGTacgtN", np.uint8)[rng.integers(0, 9, m)]
                    rows = (m + 59) // 60
                    buf = np.full((rows, 61), 10, np.uint8)
                    flat = np.zeros(rows * 60, np.uint8); flat[:m] = bases
                    buf[:, :60] = flat.reshape(rows, 60)
                    out = buf.reshape(-1)
                    if m % 60:
                        out = np.concatenate([out[:(rows - 1) * 61 + m % 60], np.array([10], np.uint8)])
                    f.write(out.tobytes())
                    done += m
    lib = capi.load_library()
    size = os.path.getsize(a.path)
    res = {"file_bytes": size, "contigs": len(lens), "bases": sum(lens)}
    for label, t in (("parallel", a.threads), ("one_thread", 1)):
        h = C.c_void_p()
        t0 = time.perf_counter()
        rc = lib.gce_fasta_load(a.path.encode(), t, C.byref(h))
        dt = time.perf_counter() - t0
        assert rc == 0
        n = C.c_int32(); ids, seqs, ln = C.POINTER(C.c_char_p)(), C.POINTER(C.c_void_p)(), C.POINTER(C.c_int64)()
        lib.gce_fasta_get(h, C.byref(n), C.byref(ids), C.byref(seqs), C.byref(ln))
        tot = sum(ln[i] for i in range(n.value))
        chk = int(np.frombuffer(C.string_at(seqs[n.value - 1], min(ln[n.value - 1], 1 << 20)), np.uint8).sum())
        lib.gce_fasta_free(h)
        res[label] = {"threads": t, "seconds": round(dt, 3), "GB_per_s": round(size / dt / 1e9, 2), "bases": tot, "tail_checksum": chk}
    from gencore_amd.shard import effective_cpus
    res["host_cpus"] = effective_cpus()
    print(json.dumps(res))
    if a.out:
        open(a.out, "w").write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
