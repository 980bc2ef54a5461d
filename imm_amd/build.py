"""Builds imm_amd/libimm_hip.so (gfx950 only) from imm_amd/csrc/*.hip with hipcc.

In-tree output: the .so is git-ignored but travels with the working tree to the GPU box.
hipcc cross-compiles without a GPU, so this runs in the CPU-only build container too.
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libimm_hip.so')
STAMP = os.path.join(HERE, '.libimm_hip.stamp')
ARCH = 'gfx950'


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _digest():
    h = hashlib.sha256()
    for p in _sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(HERE, '..', 'include', 'imm_hip.h')]:
        with open(p, 'rb') as f:
            h.update(p.encode() + b'\0' + f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [hipcc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
               '-c', src, '-o', obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
