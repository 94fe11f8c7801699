import sys, os, subprocess
sys.path[:0]=['/root/repo','/root/repo/tests']
seed=int(sys.argv[1]); mode=sys.argv[2]
os.environ["GCE_VOTE"]=mode
import ctypes as C
import fuzzgen
from gencore_amd.engine import Engine
batch, over, reference, contig_len = fuzzgen.make_case(seed)
prm=fuzzgen.make_params(over, contig_len)
e=Engine(prm)
for tid,(nib,ln) in enumerate(reference):
    if nib is not None: e.set_reference(tid,nib,ln)
e.add_reads(batch); e.finish()
e.lib.gce_debug_dump.argtypes=[C.c_void_p, C.c_char_p]
e.lib.gce_debug_dump(e._h, ("/tmp/dump_%s.txt"%mode).encode())
