"""Test-side restatement of the MX scale side-array layout (diffusionkit_amd/csrc/dk_common.h: dk_mx_scale_index), so that the
parity tests can build / read the arrays the fp8 GEMM consumes without going through the library that is under test."""
import torch


def n_blk128(rows: int) -> int:
    return (rows + 127) // 128 + 1


def scale_index(r: torch.Tensor, kb: torch.Tensor, nblk: int) -> torch.Tensor:
    """byte position of the scale of (physical row r, 32-column block kb): [kb / 4][r / 128][kb % 4][r % 16][(r / 16) % 8]"""
    return ((((kb >> 2) * nblk + (r >> 7)) * 64 + (kb & 3) * 16 + (r & 15)) << 3) + ((r >> 4) & 7)


def scales_to_array(e: torch.Tensor, rows: int = None) -> torch.Tensor:
    """E8M0 bytes [M, K / 32] -> the side array (uint8, dk_mx_scale_bytes(rows, K) bytes) of a buffer with ``rows`` rows."""
    M, KB = e.shape
    rows = M if rows is None else rows
    nblk = n_blk128(rows)
    arr = torch.zeros(((KB * 32 + 127) // 128) * nblk * 512, dtype=torch.uint8)
    r = torch.arange(M)[:, None].expand(M, KB)
    kb = torch.arange(KB)[None, :].expand(M, KB)
    arr[scale_index(r, kb, nblk).reshape(-1)] = e.reshape(-1).cpu()
    return arr


def array_to_scales(arr: torch.Tensor, M: int, K: int, rows: int = None, row0: int = 0, col0: int = 0) -> torch.Tensor:
    """the scales of rows [row0, row0 + M), columns [col0, col0 + K) of a buffer whose side array is ``arr``"""
    rows = M if rows is None else rows
    nblk = n_blk128(rows)
    KB = K // 32
    r = (row0 + torch.arange(M))[:, None].expand(M, KB)
    kb = (col0 // 32 + torch.arange(KB))[None, :].expand(M, KB)
    return arr.cpu()[scale_index(r, kb, nblk).reshape(-1)].reshape(M, KB)


def mx8_decode(q: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """e4m3 bytes [M, K] + E8M0 bytes [M, K / 32] -> fp32 values"""
    M, K = q.shape
    v = q.cpu().view(torch.float8_e4m3fn).to(torch.float32).reshape(M, K // 32, 32)
    return (v * torch.ldexp(torch.ones(M, K // 32, 1), e.cpu().to(torch.int32)[..., None] - 127)).reshape(M, K)
