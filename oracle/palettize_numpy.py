"""TEST INFRASTRUCTURE -- the checker for palettized (CCV_QX) tensors: nnc_mi355x_depalettize and the rows that take palettized inputs (ccv_amd/csrc/palette.cpp);
never imported by the product path (only tests/ import it).

PARITY PINNED.  This file restates, in numpy, the byte layout the reference's quantiser writes and its CPU reader expands:
  ccv_nnc_palettize      lib/nnc/ccv_nnc_palettize.c:9-208     (per block: k-means palette of 2^qbits values, then the indices)
  _ccv_nnc_depalettize   lib/nnc/ccv_nnc_palettize.c:211-956   (block stride = palette + number_in_blocks / 8 * qbits ... index bytes -- the integer divisions of
                                                                 each bit width --; indices a big-endian bit stream, element j of a block in bits [j q, (j + 1) q))
  tensor size            lib/nnc/ccv_nnc_easy.h:220-238        (ccv_nnc_tensor_data_size_without_padding for CCV_QX)
The restatement exists because the quantiser only ever emits k-means palettes: `pack_stream` builds the stream for ANY palette and ANY indices, so that the kernels
are checked on full-range random words.  It is pinned both ways against the reference itself (oracle/_ref/libccv_ref.so exports both functions): the reference's
reader expands pack_stream's bytes to expand()'s values, and the reference's quantiser writes streams this layout reads back to the values that went in
(tests/test_palettize.py::test_the_numpy_stream_is_what_the_reference_reader_expands, ::test_the_reference_quantiser_writes_the_numpy_stream) -- the cases include the
reference's own test sizes (test/int/nnc/palettize.tests.c: 2839 / 2840 / 8192 elements, 128 / 512 / 1280 per block)."""
import numpy as np

CCV_32F, CCV_64F, CCV_16F = 0x04000, 0x10000, 0x20000  # lib/ccv.h:46-51

WORD = {CCV_16F: np.uint16, CCV_32F: np.uint32, CCV_64F: np.uint64}  # values are moved, not interpreted: compare the words


def index_bytes_per_block(qbits, nib):
    return {4: nib // 2, 5: nib // 8 * 5, 6: nib // 4 * 3, 7: nib // 8 * 7, 8: nib}[qbits]


def pack_stream(palettes, indices, qbits, nib, datatype):
    """palettes: [blocks][2^qbits] words, indices: [count] ints < 2^qbits.  The stream ccv_nnc_palettize would have written for these choices:
    per block the palette, then its elements' indices as a big-endian bit stream; the last block's indices end with its last whole group."""
    count = len(indices)
    blocks = (count + nib - 1) // nib
    word = np.dtype(WORD[datatype])
    group = {4: 2, 5: 8, 6: 4, 7: 8, 8: 1}[qbits]
    out = bytearray()
    for b in range(blocks):
        out += np.asarray(palettes[b], dtype=word).tobytes()
        idx = np.asarray(indices[b * nib:(b + 1) * nib], dtype=np.uint64)
        full = len(idx) == nib
        padded = (len(idx) + group - 1) // group * group
        idx = np.concatenate([idx, np.zeros(padded - len(idx), np.uint64)])
        bits = ((idx[:, None] >> np.arange(qbits - 1, -1, -1, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8).reshape(-1)
        packed = np.packbits(bits).tobytes()
        if full:
            assert len(packed) >= index_bytes_per_block(qbits, nib)
            packed = packed[:index_bytes_per_block(qbits, nib)] + bytes(max(0, index_bytes_per_block(qbits, nib) - len(packed)))
        out += packed
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def expand(palettes, indices, nib, datatype):
    idx = np.asarray(indices)
    pal = np.asarray(palettes, dtype=WORD[datatype])
    return pal[np.arange(len(idx)) // nib, idx]
