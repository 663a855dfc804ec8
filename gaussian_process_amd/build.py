"""Build libgpmi355x.so in-tree:  python -m gaussian_process_amd.build [--clean]"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def build(clean=False, jobs=6, quiet=False):
    if clean:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if out.returncode != 0:
        sys.stderr.write(out.stdout)
        raise RuntimeError("building libgpmi355x.so failed (exit %d)" % out.returncode)
    if not quiet:
        sys.stdout.write(out.stdout[-2000:])
    return os.path.join(os.path.dirname(CSRC), "libgpmi355x.so")


if __name__ == "__main__":
    print(build(clean="--clean" in sys.argv))
