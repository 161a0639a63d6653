#!/usr/bin/env python3
"""Yardstick for the approximate GEMM's ceiling: what does the VENDOR int8 GEMM (hipBLASLt behind torch._int_mm,
int8 x int8 -> int32) deliver on this box at the block's shape — 11648 x 11648 x 5120, the row lists of a 10k x 10k C4 block — and at a
span's shape (7 blocks on the to side)?  Tools only: never linked into or called by the product.  The product's kernel
(gemm_apx_kernel) feeds its MFMAs from BIT-packed operands expanded in registers (15 MB of panels per block instead of 120 MB of
bytes); this is the byte-operand feed it is compared with in docs/HISTORY.md 5.1c(b).
usage: python tools/i8_gemm_yardstick.py [out.json]"""
import json
import sys
import time

import torch


def bench(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters


def main():
    out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "peak_i8_TOPs": 5000.0, "peak_bf16_TFLOPs": 2500.0, "cases": []}
    dev = torch.device("cuda", 0)
    for name, (M, N, K) in {"block 10k x 10k (row lists 11648 x 11648, K = 5120)": (11648, 11648, 5120),
                            "span of 7 blocks (81536 x 11648, K = 5120)": (81536, 11648, 5120),
                            "square 8192^3": (8192, 8192, 8192)}.items():
        rec = {"case": name, "M": M, "N": N, "K": K}
        a = torch.randint(-127, 127, (M, K), dtype=torch.int8, device=dev)
        b = torch.randint(-127, 127, (K, N), dtype=torch.int8, device=dev)
        try:
            ms = bench(lambda: torch._int_mm(a, b))
            rec["int8_ms"] = ms
            rec["int8_TOPs"] = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            rec["int8_frac_of_peak"] = rec["int8_TOPs"] / out["peak_i8_TOPs"]
        except Exception as e:   # noqa: BLE001
            rec["int8_error"] = repr(e)[:300]
        del a, b
        x = torch.randn((M, K), dtype=torch.bfloat16, device=dev)
        y = torch.randn((K, N), dtype=torch.bfloat16, device=dev)
        ms = bench(lambda: torch.matmul(x, y))
        rec["bf16_ms"] = ms
        rec["bf16_TFLOPs"] = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        rec["bf16_frac_of_peak"] = rec["bf16_TFLOPs"] / out["peak_bf16_TFLOPs"]
        del x, y
        out["cases"].append(rec)
        print(json.dumps(rec), flush=True)
    out["when"] = time.strftime("%Y-%m-%d %H:%M:%S")
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
