"""Worker of test_hip_accelerator.test_rccl_entry_points_of_the_c_abi_in_a_group_of_one (a child process with a time limit: an RCCL
bring-up problem must not hang the test process).  The C-ABI's own communicator (include/xfm_hip.h xfm_dp_*, wrapped by xfm_amd/dp.py)
with ONE rank on cuda:0: every collective is then the identity, so what comes back must be what went in, bit for bit -- fp32, bf16 and
int32; sum, mean, max; gather; broadcast; on a side stream, ordered by stream semantics only.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from xfm_amd.dp import ID_BYTES, NativeComm  # noqa: E402


def main():
    torch.cuda.set_device(0)
    uid = NativeComm.unique_id()
    assert len(uid) == ID_BYTES and any(uid)
    comm = NativeComm(uid, 0, 1)
    g = torch.Generator(device="cuda").manual_seed(5)
    out = {"bad": 0, "checks": 0}

    def same(a, b, what):
        out["checks"] += 1
        if not torch.equal(a, b):
            out["bad"] += 1
            out.setdefault("which", []).append(what)

    side = torch.cuda.Stream()
    for dtype in (torch.float32, torch.bfloat16):
        for op in ("sum", "avg", "max"):
            x = torch.randn(1_000_003 if dtype == torch.float32 else 4096, device="cuda", generator=g).to(dtype)
            want = x.clone()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):     # (the accelerator's exchange runs on its communication stream)
                comm.all_reduce(x, op)
            torch.cuda.current_stream().wait_stream(side)
            same(x, want, f"all_reduce {dtype} {op}")
    feat = torch.randn(64, 256, device="cuda", generator=g)
    same(comm.all_gather(feat), feat, "all_gather")
    ids = torch.arange(77, device="cuda", dtype=torch.int32)
    same(comm.all_gather(ids), ids, "all_gather int32")
    p = torch.randn(123_457, device="cuda", generator=g)
    want = p.clone()
    same(comm.broadcast(p, 0), want, "broadcast")
    same(comm.all_reduce(torch.empty(0, device="cuda")), torch.empty(0, device="cuda"), "empty bucket")
    torch.cuda.synchronize()
    comm.finalize()
    comm.finalize()   # (idempotent)
    print("DP_W1 " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
