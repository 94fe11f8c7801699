"""Depth / BED statistics: Stats::statDepth + Bed::statDepth (src/stats.cpp:57-84, src/bed.cpp:66-81) and the BED
loader (src/bed.cpp:111-168).  CPU: the oracle's restatement against hand-worked cases, the loader against hand-worked files.
GPU: gce_depth_stats against the oracle on synthetic streams with a BED panel."""
import numpy as np
import pytest

from gencore_amd.capi import default_params


def test_stat_depth_by_hand(oracle):
    """step 100, contig of 450 bp -> 1 + 450/100 = 5 bins (stats.cpp:41-47)."""
    L = oracle.lib()
    d = np.zeros(5, np.int64)
    L.orc_stat_depth(d.ctypes.data, 5, 100, 90, 150)      # [90, 240): 10 in bin 0, all of bin 1, 40 in bin 2
    assert d.tolist() == [10, 100, 40, 0, 0]
    L.orc_stat_depth(d.ctypes.data, 5, 100, 120, 30)      # inside bin 1: + len
    assert d.tolist() == [10, 130, 40, 0, 0]
    L.orc_stat_depth(d.ctypes.data, 5, 100, 380, 20)      # end = 400 -> rightPos 4 != leftPos 3: 20 to bin 3, 0 to bin 4
    assert d.tolist() == [10, 130, 40, 20, 0]
    L.orc_stat_depth(d.ctypes.data, 5, 100, 450, 60)      # end = 510 -> rightPos 5 >= 5 bins: the whole read is dropped (stats.cpp:69-70)
    assert d.tolist() == [10, 130, 40, 20, 0]


def test_bed_depth_by_hand(oracle):
    L = oracle.lib()
    rs, re_ = np.asarray([100, 150, 300, 50], np.int32), np.asarray([200, 400, 350, 60], np.int32)      # the last region is out of order
    cnt = np.zeros(4, np.int64)
    L.orc_bed_depth(rs.ctypes.data, re_.ctypes.data, cnt.ctypes.data, 4, 180, 50)                         # read [180, 230]
    # region 0: min(200,230) - max(100,180) = 20; region 1: 230 - 180 = 50; region 2 starts at 300 > 230: break -- region 3 is never seen
    assert cnt.tolist() == [20, 50, 0, 0]
    L.orc_bed_depth(rs.ctypes.data, re_.ctypes.data, cnt.ctypes.data, 4, 200, 10)                         # touching region 0 at its end: + 0, region 1: + 10
    assert cnt.tolist() == [20, 60, 0, 0]
    L.orc_bed_depth(rs.ctypes.data, re_.ctypes.data, cnt.ctypes.data, 4, 20, 35)                          # [20, 55]: regions 0..2 start beyond 55 -> break at once,
    assert cnt.tolist() == [20, 60, 0, 0]                                                                 # the unsorted region [50, 60] is missed (bed.cpp:75-76)


BED_CASES = [
    # (text, target names, expected [(tid, start, end, name)])
    ("chr1\t100\t200\tgeneA\nchr2\t5\t50\n# comment\tx\ty\nchrUn\t1\t2\tn\nchr1\t300\t400\t  padded  \n", ["chr1", "chr2"],
     [(0, 100, 200, "geneA"), (1, 5, 50, ""), (-1, 1, 2, "n"), (0, 300, 400, "padded")]),
    ("chr1\t10\t20\r\n\r\nchr1\t30\n  chr1\t40\t50\textra\tcols\n", ["chr1"], [(0, 10, 20, ""), (0, 40, 50, "extra")]),
    ("chr1\t1\t2\n" + "x" * 5000 + "\nchr1\t3\t4\n", ["chr1"], [(0, 1, 2, "")]),          # a line that does not fit getline's buffer ends the loop
    ("dup\t1x\t-7abc\n", ["dup", "dup"], [(1, 1, -7, "")]),                                # atoi; the LAST header contig of that name wins
]


@pytest.mark.parametrize("case", range(len(BED_CASES)))
def test_bed_loader(built, tmp_path, case):
    from gencore_amd.bamio import load_bed
    text, names, want = BED_CASES[case]
    p = tmp_path / "x.bed"
    p.write_bytes(text.encode())
    assert load_bed(str(p), names) == want


def depth_case(workload, n_pairs, shuffle_bed=False):
    from gencore_amd import synth
    d = synth.generate(workload, n_pairs=n_pairs)
    batch = d.to_batch()
    tl = np.asarray(d.target_len, np.uint32)
    prm = default_params(n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=d.info["umi_prefix"], cluster_size_req=d.info["supporting_reads"])
    prm._keep = tl
    bed = d.info.get("bed")
    if bed is None:                                            # no panel in this workload: tile the contigs, overlapping regions included
        rng = np.random.RandomState(3)
        regs = []
        for t, ln in enumerate(tl):
            st = np.sort(rng.randint(0, max(int(ln) - 500, 1), 40))
            regs += [(t, int(a), int(a) + int(rng.randint(50, 3000))) for a in st]
        bed = np.asarray(regs, np.int64)
    regions = [tuple(int(x) for x in r) for r in np.asarray(bed)[:, :3]]
    if shuffle_bed:                                            # an unsorted BED: the literal loop with its early break must be reproduced
        rng = np.random.RandomState(5)
        regions = [regions[i] for i in rng.permutation(len(regions))]
    return d, batch, prm, regions


@pytest.mark.gpu
@pytest.mark.parametrize("workload,n_pairs,step,shuffle", [("cfg3", 30000, 10000, False), ("cfg2", 20000, 1000, False), ("cfg3", 8000, 10000, True), ("cfg5", 3000, 250, False)])
def test_depth_stats_on_the_engine(built, oracle, workload, n_pairs, step, shuffle):
    from gencore_amd.engine import Engine
    d, batch, prm, regions = depth_case(workload, n_pairs, shuffle)
    if shuffle:
        regions = regions[:60]
    want_t = oracle.run(batch, prm, d.reference_host())
    off, pre_d, post_d, pre_b, post_b = oracle.depth_stats(batch, want_t, d.target_len, step, regions)
    eng = Engine(prm)
    try:
        for tid, (nib, ln) in enumerate(d.reference_host()):
            if nib is not None:
                eng.set_reference(tid, nib, ln)
        eng.add_reads(batch)
        eng.finish()
        g_off, g_pre_d, g_post_d, g_pre_b, g_post_b = eng.depth_stats(step, regions)
    finally:
        eng.close()
    assert np.array_equal(g_off, off)
    assert np.array_equal(g_pre_d, pre_d) and np.array_equal(g_post_d, post_d)
    assert np.array_equal(g_pre_b, pre_b) and np.array_equal(g_post_b, post_b)
    assert pre_d.sum() > post_d.sum() > 0 and (workload != "cfg3" or pre_b.sum() > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,n_pairs,shards,mode,step", [("cfg3", 30000, 1, 0, 10000), ("cfg3", 30000, 3, 0, 10000), ("cfg2", 20000, 4, 0, 1000), ("cfg5", 3000, 2, 1, 250)])
def test_file_runners_carry_the_depth_statistics(built, oracle, tmp_path, workload, n_pairs, shards, mode, step):
    """gce_run_bam_depth: the file runners (one engine, and the sharded runner on the GPU codec: several engines on device 0 here) with Options::coverageStep and
    a BED file.  Every engine adds up its own reads and records on its GPU (gce_stats_payload_device), the payloads -- Stats blocks + depth bins + region counts in
    ONE buffer -- are summed in device memory (gce_stats_payload_sum).  Depth bins and region counts equal the oracle's over the whole stream
    (src/stats.cpp:56-83, src/bed.cpp:64-79), the Stats blocks inside the payload equal the runner's own."""
    import pybam
    from gencore_amd.bamio import run_bam_depth
    from test_bamio import records_of
    d, batch, prm0, regions = depth_case(workload, n_pairs)
    tl = np.asarray(d.target_len, np.uint32)
    targets = [("chr%d" % (i + 1), int(l)) for i, l in enumerate(tl)]
    src, dst, bed = (str(tmp_path / x) for x in ("in.bam", "out.bam", "panel.bed"))
    pybam.write_bam(src, records_of(batch), targets)
    with open(bed, "w") as f:
        f.write("# panel\n")
        for t, a, z in regions:
            f.write("chr%d\t%d\t%d\tr\n" % (t + 1, a, z))
        f.write("chrNotInHeader\t5\t50\tx\n")                                    # kept in the list with tid -1 (bed.cpp:151-166 drops it from the map): counts 0
    want_t = oracle.run(batch, prm0, [])                                         # (no FASTA in either run)
    off, pre_d, post_d, pre_b, post_b = oracle.depth_stats(batch, want_t, d.target_len, step, regions)
    prm = default_params(umi_prefix="auto", cluster_size_req=d.info["supporting_reads"])
    run, got = run_bam_depth(src, dst, prm, [0] * shards, step, bed=bed, plan_mode=mode, threads=4)
    assert run.n_out > 0 and got["regions"][:len(regions)] == regions and got["regions"][-1][0] == -1
    assert np.array_equal(got["bin_off"], off)
    assert np.array_equal(got["pre_depth"], pre_d) and np.array_equal(got["post_depth"], post_d)
    assert np.array_equal(got["pre_bed"][:-1], pre_b) and np.array_equal(got["post_bed"][:-1], post_b) and got["pre_bed"][-1] == 0 and got["post_bed"][-1] == 0
    assert got["pre"] == bytes(run.pre) and got["post"] == bytes(run.post)
    assert got["pre"] == bytes(want_t.pre) and got["post"] == bytes(want_t.post)
    assert got["payload_bytes"] == 8 * (2 * 114 + 2 * len(pre_d) + 2 * (len(regions) + 1))
