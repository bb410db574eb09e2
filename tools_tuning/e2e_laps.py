"""The bench's end_to_end leg by hand: the juliet front end as a child process on a 100k-read rich-QV BAM, five runs, the wall time by
this process's clock and the child's own stage laps of every run.   (on the GPU box)"""
import sys, os, re, subprocess, time, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
bindir = os.path.join(ROOT, "minorseq_amd", "bin")
tmp = tempfile.mkdtemp(prefix="jl_e2e_")
bam, cfg, outj = os.path.join(tmp, "in.bam"), os.path.join(tmp, "cfg.json"), os.path.join(tmp, "out.json")
subprocess.check_call([os.path.join(bindir, "juliet-synth"), "--reads", "100000", "--cols", "3000", "--seed", "1000", "--ref-seed", "2", "--rich-qv", "-o", bam, "--config-out", cfg])
cmd = [os.path.join(bindir, "juliet"), "--timing", "-c", cfg, "--mode-phasing", "--min-qv", "20", bam, outj]
for i in range(int(os.environ.get("RUNS", "5"))):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    wall = 1000 * (time.perf_counter() - t0)
    print("wall %.1f" % wall)
    for ln in p.stderr.decode().splitlines():
        if "timing" in ln and ("(at" in ln): print("   ", ln.replace("juliet: timing ", ""))
