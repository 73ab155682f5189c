"""cfg2 through the seqalign tool: FASTA in -> HIP alignment -> HDF5 out, with the -B phase report
(development helper, run through gpurun)."""
import sys, pathlib, subprocess, tempfile, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tests.synth import make_config
ROOT = pathlib.Path(__file__).resolve().parents[1]
seqs, cfg = make_config("cfg2")
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    fa, out = pathlib.Path(d) / "cfg2.fasta", pathlib.Path(d) / "cfg2.h5"
    fa.write_bytes(b"".join(b">s%d\n" % k + s + b"\n" for k, s in enumerate(seqs)))
    for extra in ([],):
        t = time.time()
        r = subprocess.run([str(ROOT / "cli" / "seqalign"), "-i", str(fa), "-o", str(out), "-a", "nw", "-m", "blosum62", "-p", "4", "-F", "-B", *extra],
                           capture_output=True, text=True, timeout=600)
        print("flags", extra, "rc", r.returncode, "wall %.2f s" % (time.time() - t))
        print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if any(w in l for w in ("second", "time", "Time", "took", "GCUPS", "rror"))))
        if out.exists():
            print("hdf5 bytes", out.stat().st_size); out.unlink()
