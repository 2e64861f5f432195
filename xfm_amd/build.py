"""Build libxfm_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libxfm_hip.so")
def _sources():
    """Every file the single translation unit (capi.hip) can include: whatever sits in csrc/ plus the public header -- listed from the
    directory, so a new kernel file cannot be forgotten here (round 4: attention_long.hip was, and is_stale() missed edits to it)."""
    files = [f for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    return files + [os.path.join("..", "..", "include", "xfm_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in _sources())


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-inline-asm"]


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc()] + FLAGS + ["-shared", os.path.join(CSRC, "capi.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB


_DEVICE_COMPILE = {}


def _device_compile():
    """One device-only compile to assembly with -Rpass-analysis=kernel-resource-usage (no GPU needed) -> (remarks, assembly text);
    cached for the process (40 s)."""
    if "asm" not in _DEVICE_COMPILE:
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "xfm.s")
            cmd = [_hipcc()] + FLAGS + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, "capi.hip"),
                                        "-o", out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
            _DEVICE_COMPILE["remarks"] = r.stderr
            with open(out) as f:
                _DEVICE_COMPILE["asm"] = f.read()
    return _DEVICE_COMPILE["remarks"], _DEVICE_COMPILE["asm"]


def resource_usage():
    """{kernel symbol: {"vgprs": n, "agprs": n, "spill": n, "scratch": bytes per lane, "lds": bytes}} from a device-only compile with
    -Rpass-analysis=kernel-resource-usage (no GPU needed): the register-bound kernels (256 x 256 GEMM tiles: two waves per SIMD,
    <= 256 VGPRs, no scratch) are held to their budgets by tests/test_capi.py."""
    import re
    remarks, _ = _device_compile()
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "VGPRs Spill": "spill", "ScratchSize [bytes/lane]": "scratch", "LDS Size [bytes/block]": "lds",
            "SGPRs Spill": "sgpr_spill"}
    for line in remarks.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark: +([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def kernel_isa():
    """{kernel symbol: [instruction lines]} of the gfx950 code (same compile as resource_usage): what tests/test_capi.py reads to pin
    the instruction counts that the counted `s_waitcnt vmcnt` schedules depend on."""
    import re
    _, asm = _device_compile()
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            s = line.strip()
            if s and not s.startswith((";", ".")) :
                cur.append(s)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
