#!/usr/bin/env python
"""One-line digest of a bench.py JSON line (and of its other_configs legs): scripts/bench_line.py <file.json>"""
import json
import sys


def digest(tag, d):
    r = d.get("roofline") or {}
    att, conv = r.get("attention") or {}, r.get("conv") or {}
    print(f"{tag}: {d['value']} images/s, {d['ms_per_step']} ms/image; denoise {d.get('denoise_ms_per_step')} ms/step, decode {d.get('vae_decode_ms')} ms; "
          f"{r.get('kernel', 'gemm')} {r.get('achieved')} {r.get('unit')} = {r.get('frac')}; attention {att.get('achieved')}; conv {conv.get('achieved')}; "
          f"whole path {d.get('mfma_roofline_frac_whole_path')}")


def main():
    lines = [ln for ln in open(sys.argv[1]).read().strip().splitlines() if ln.startswith("{")]
    if not lines:
        print(sys.argv[1], ": no JSON line")
        return
    d = json.loads(lines[-1])
    digest(sys.argv[1].split("/")[-1], d)
    for k, v in (d.get("other_configs") or {}).items():
        digest("  " + k.split(" (")[0], v)
    cb = d.get("cpu_baseline")
    if cb:
        print(f"  cpu_baseline: {cb['value']:.5f} {cb['unit']} on {cb['cores']} threads ({cb.get('sample_seconds')} s sample)")


if __name__ == "__main__":
    main()
