"""Builds imm_amd/libimm_hip.so (gfx950 only) from imm_amd/csrc/*.hip with hipcc.

In-tree output: the .so is git-ignored but travels with the working tree to the GPU box.
hipcc cross-compiles without a GPU, so this runs in the CPU-only build container too.
The library carries the sha256 of its sources (imm_source_digest(), repo-relative paths + contents); a rebuild is
skipped only when the existing binary carries the digest of the current checkout — there is no stamp file to go stale.
"""
import fcntl
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libimm_hip.so')
ARCH = 'gfx950'


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _extra_flags():
    # IMM_HIPCC_FLAGS: extra compile flags for diagnosis builds (e.g. -DIMM_HDEEP_PROFILE), never set in normal use
    return os.environ.get('IMM_HIPCC_FLAGS', '').split()


def source_digest():
    """sha256 over the sources AND what they are compiled with (target arch, extra flags): a diagnosis build never passes for
    the production library."""
    h = hashlib.sha256()
    h.update(('arch=%s flags=%s\0' % (ARCH, ' '.join(_extra_flags()))).encode())
    for p in _sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(ROOT, 'include', 'imm_hip.h')]:
        with open(p, 'rb') as f:
            h.update(os.path.relpath(p, ROOT).replace(os.sep, '/').encode() + b'\0' + f.read() + b'\0')
    return h.hexdigest()


def library_digest(path=LIB):
    """Digest embedded in a built library, read from the file's bytes (no dlopen); '' when absent."""
    try:
        with open(path, 'rb') as f:
            blob = f.read()
    except OSError:
        return ''
    i = blob.find(b'IMM_SOURCE_DIGEST=')
    if i < 0:
        return ''
    d = blob[i + 18:i + 18 + 64]
    return d.decode('ascii', 'replace') if len(d) == 64 and all(c in b'0123456789abcdef' for c in d) else ''


def have_compiler():
    return os.path.exists(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'))


def build(force=False, verbose=True):
    dig = source_digest()
    if not force and library_digest() == dig:
        return LIB
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    # One builder at a time: N ranks started together on a stale library (torchrun, bench.py --gpus N) would otherwise all
    # run hipcc over the same object files and could link each other's half-written objects.  The others wait on the lock
    # and then find the library current.
    with open(os.path.join(objdir, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and library_digest() == dig:
                return LIB
            return _build_locked(dig, objdir, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(dig, objdir, verbose):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [hipcc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
               '-DIMM_SOURCE_DIGEST="%s"' % dig] + _extra_flags() + ['-c', src, '-o', obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    tmp = LIB + '.tmp.%d' % os.getpid()
    subprocess.check_call([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', tmp] + objs)
    os.replace(tmp, LIB)     # a process that has the old file mapped keeps its inode
    return LIB


def build_variant(out, flags):
    """A/B builds: the same sources with extra compile flags into another file (objects under build/<name>/); load it with
    IMM_HIP_LIB=<out>."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(HERE, 'build', os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        procs.append(subprocess.Popen([hipcc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-DIMM_SOURCE_DIGEST="variant"'] +
                                      flags + ['-c', src, '-o', obj]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed')
    subprocess.check_call([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', out] + objs)
    return out


if __name__ == '__main__':
    if '--out' in sys.argv:
        i = sys.argv.index('--out')
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build(force='--force' in sys.argv))
