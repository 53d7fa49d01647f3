"""Per-stage table of one VAE decode: FLOP, microseconds and TFLOP/s of every convolution launch, in launch order.

    cd /tmp && rocprofv3 --kernel-trace -d <out> -o vae --output-format csv -- python scripts/vae_decode_bench.py     (N=3 is enough)
    python scripts/vae_stage_table.py <out>            -> markdown on stdout (profiles/r05_vae_by_stage.md)

The decoder's launch order is fixed (vae.py:336-401; diffusionkit_amd/csrc/engine.hip: dk_vae_decode): conv_in, mid resnet, mid attention,
mid resnet, then the up blocks from the deepest (3 resnets = 6 convs each; an upsampling conv behind all but the last), conv_out.  The
fused stages run dk_conv_halo_kernel<128> (GroupNorm-apply + SiLU on the way into the LDS halo tile; the 1x1 shortcut of a
channel-changing resnet rides in its conv2 as extra K-tiles) or, with >= 256 output channels where one image fills the CUs,
dk_conv256v4_kernel (one wave per SIMD, asm body); conv_out dk_conv_halo_kernel<16, true> (+ clip + uint8).
"""
import csv
import glob
import os
import sys


def stages(latent=128, chans=(128, 256, 512, 512), layers=3):
    """(name, H, Cin, Cout, extra_k) of the 31 conv_halo<128> launches in order; extra_k = shortcut channels folded into the reduction"""
    out = []
    H = latent
    Cm = chans[-1]
    for r in (0, 2):
        out.append((f"mid_blocks.{r}.conv1", H, Cm, Cm, 0))
        out.append((f"mid_blocks.{r}.conv2", H, Cm, Cm, 0))
    C = Cm
    n = len(chans)
    for j in range(n - 1, -1, -1):
        Cout = chans[j]
        for r in range(layers):
            cin = C if r == 0 else Cout
            out.append((f"up_blocks.{j}.resnets.{r}.conv1", H, cin, Cout, 0))
            out.append((f"up_blocks.{j}.resnets.{r}.conv2", H, Cout, Cout, cin if cin != Cout else 0))
        C = Cout
        if j > 0:
            H *= 2
            out.append((f"up_blocks.{j}.upsample.conv (nearest x2 view)", H, C, C, 0))
    return out


def main():
    root = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    conv = [(s, e, k) for s, e, k in rows if "dk_conv_halo_kernel<128" in k or "dk_conv256v4_kernel" in k]
    st = stages()
    n = len(st)
    assert len(conv) % n == 0 and conv, f"{len(conv)} conv_halo<128> launches, expected a multiple of {n}"
    decodes = len(conv) // n
    last = conv[-n:]
    # whole decode: first kernel after the previous decode's last conv_out .. this decode's conv_out
    outk = [(s, e) for s, e, k in rows if "dk_conv_halo_kernel<16" in k]
    t_end = outk[-1][1]
    t_begin = outk[-2][1] if len(outk) > 1 else rows[0][0]
    inside = [(s, e, k) for s, e, k in rows if s >= t_begin and e <= t_end]
    busy = sum(e - s for s, e, k in inside) / 1e3
    print(f"# VAE decode 128 x 128 latent -> 1024 x 1024, per launch (last of {decodes} decodes in the trace)\n")
    print(f"whole decode: {(t_end - inside[0][0]) / 1e3:.0f} us wall, kernels {busy:.0f} us ({len(inside)} launches)\n")
    print("| # | stage | pixels | C_in -> C_out | GFLOP | us | TFLOP/s | of 2500 | kernel |\n|---|---|---|---|---|---|---|---|---|")
    tot_f = tot_t = 0.0
    by_res = {}
    for i, ((name, H, cin, cout, xk), (s, e, kn)) in enumerate(zip(st, last)):
        fl = 2.0 * H * H * (9 * cin + xk) * cout
        us = (e - s) / 1e3
        tot_f += fl
        tot_t += us
        key = (H, cin, cout)
        by_res.setdefault(key, [0.0, 0.0, 0])
        by_res[key][0] += fl
        by_res[key][1] += us
        by_res[key][2] += 1
        print(f"| {i + 1} | `{name}` | {H}² | {cin}{' (+' + str(xk) + ' shortcut)' if xk else ''} -> {cout} | {fl / 1e9:.0f} | {us:.1f} | {fl / us / 1e6:.0f} | {fl / us / 1e6 / 2500:.3f} | {'conv256v4' if 'conv256v4' in kn else 'conv_halo'} |")
    print(f"| | **all 31** | | | {tot_f / 1e9:.0f} | {tot_t:.0f} | {tot_f / tot_t / 1e6:.0f} | {tot_f / tot_t / 1e6 / 2500:.3f} |")
    print("\n| shape class | launches | GFLOP | us | TFLOP/s |\n|---|---|---|---|---|")
    for (H, cin, cout), (fl, us, cnt) in sorted(by_res.items()):
        print(f"| {H}² {cin} -> {cout} | {cnt} | {fl / 1e9:.0f} | {us:.0f} | {fl / us / 1e6:.0f} |")
    other = {}
    for s, e, k in inside:
        if "dk_conv_halo_kernel<128" in k or "dk_conv256v4_kernel" in k:
            continue
        kk = k.split("(")[0].replace("void ", "")
        other.setdefault(kk, [0.0, 0])
        other[kk][0] += (e - s) / 1e3
        other[kk][1] += 1
    print("\n| other kernels of the decode | launches | us |\n|---|---|---|")
    for k, (us, cnt) in sorted(other.items(), key=lambda kv: -kv[1][0]):
        print(f"| `{k[:90]}` | {cnt} | {us:.1f} |")


if __name__ == "__main__":
    main()
