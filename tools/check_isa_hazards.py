"""Static check of the built gfx950 code for a store-data hazard the compiler does not guard.

Measured on MI355X (tools/debug history): `buffer_store_dwordx4 v[a:a+3], v, s[..], sN offen` -- a wide buffer
store whose soffset is an SGPR -- followed directly by a VALU write of v[a:a+3] stores garbage in the lanes whose data is read
last.  LLVM's hazard recognizer adds the wait state only when soffset is not a register.  The kernels therefore never put an
SGPR soffset on a wide store; this script proves it on the shipped binary (and lists any offender).

    python tools/check_isa_hazards.py [path/to/libctl_hip.so]      exit code 1 if a hazard candidate exists
"""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "cooperative_training_and_latent_space_data_augmentation_amd", "csrc", "libctl_hip.so")
WIDE_STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*\w+,\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi|ttmp\d+)\b")


def device_objects(so_path, workdir):
    local = os.path.join(workdir, os.path.basename(so_path))
    shutil.copy(so_path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=workdir)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn-amd-amdhsa--gfx950" in f)


def scan(so_path=DEFAULT):
    """Returns (number of kernels scanned, number of wide buffer stores seen, [offending 'kernel: instruction' strings])."""
    offenders, kernels, stores = [], 0, 0
    with tempfile.TemporaryDirectory() as wd:
        objs = device_objects(so_path, wd)
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {so_path}")
        for obj in objs:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], check=True,
                                 capture_output=True, text=True).stdout
            kernel = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    kernel = m.group(1)
                    kernels += 1
                    continue
                if "buffer_store_dwordx" in line:
                    stores += 1
                    if WIDE_STORE.match(line):
                        offenders.append(f"{kernel}: {line.strip().split('//')[0].strip()}")
    return kernels, stores, offenders


if __name__ == "__main__":
    k, s, off = scan(sys.argv[1] if len(sys.argv) > 1 else DEFAULT)
    print(f"{k} kernels, {s} buffer stores, {len(off)} wide stores with an SGPR soffset")
    for o in off[:20]:
        print("  ", o)
    sys.exit(1 if off else 0)
