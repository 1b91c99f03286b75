// Thin C-ABI driver around the UNMODIFIED reference sources, compiled where they
// lie (-I/root/reference/PreProcessing) into oracle/_ref/libref.so by
// oracle/Makefile.  TEST INFRASTRUCTURE ONLY: it exists to pin oracle/spmm_oracle.cpp
// and to generate the fixtures under tests/golden/ (tools/gen_fixtures.py).
// No reference source text is copied: the translation unit includes the
// reference's .cpp and calls its functions.
//
// The reference only prints a time; the arrays it computes are handed out here
// by calling its functions directly.  Per-block arrays never leave
// gen_trace_formats (serial_newblock_clock.cpp:289-307), so ref_stage_block()
// re-runs one block with the reference's own per-block functions
// (gen_panel_list v8sort.h:49, panel_sort_nnz v8sort.h:152, reorder_row
// transmat.h:135, transpose_spv8_nnz v8sort.h:300, gen_serial_origin_vec
// serial_newblock_clock.cpp:187).
#define main ref_main
#include "serial_newblock_clock.cpp"
#undef main

#include <cstdint>

static SpM* g_mat = nullptr;

extern "C" {

// csr_matrix (serial_newblock_clock.cpp:47): load, keep, report sizes.
int ref_load(const char* path, int* shape3) {
    delete g_mat;
    g_mat = new SpM();
    *g_mat = csr_matrix(std::string(path));
    for (int i = 0; i < 3; i++) shape3[i] = g_mat->shape[i];
    return 0;
}

int ref_get_csr(int* indptr, int* indices, double* data) {
    if (!g_mat) return -1;
    std::memcpy(indptr, g_mat->indptr, sizeof(int) * (g_mat->shape[0] + 1));
    std::memcpy(indices, g_mat->indices, sizeof(int) * g_mat->shape[2]);
    std::memcpy(data, g_mat->data, sizeof(double) * g_mat->shape[2]);
    return 0;
}

// Build the kept matrix from arrays instead of a file.
int ref_set_csr(int nrow, int ncol, int nnz, int* indptr, int* indices, double* data) {
    delete g_mat;
    int shape[3] = {nrow, ncol, nnz};
    g_mat = new SpM(data, indices, indptr, shape);
    return 0;
}

// gen_trace_formats (serial_newblock_clock.cpp:221) on the kept matrix.
// seq_input[nnz], rseq[nrow], bsize_list[1000], seq_offset[1000] are caller buffers
// of the sizes main() uses (serial_newblock_clock.cpp:543-553).
int ref_trace_formats(int* seq_input, int* end_seq, int* rseq, int* bnum, int* bsize_list,
                      int* seq_offset) {
    if (!g_mat) return -1;
    std::vector<int> rs;
    int** bc = nullptr; double** bd = nullptr; int** bi = nullptr;
    int shape[3];
    int e = 0, bn = 0;
    int* seq_ref = seq_input;
    gen_trace_formats(*g_mat, seq_ref, e, rs, bn, bc, bd, bi, shape, bsize_list, seq_offset);
    *end_seq = e;
    *bnum = bn;
    for (size_t i = 0; i < rs.size(); i++) rseq[i] = rs[i];
    return (int)rs.size();
}

// bitmap_reorder (bitmap.h:108) on the kept matrix; the result REPLACES a second
// kept matrix used by the block-level calls below.
static SpM* g_reord = nullptr;
int ref_bitmap_reorder(int sect, int* seq, int* indptr, int* indices, double* data) {
    if (!g_mat) return -1;
    delete g_reord;
    g_reord = new SpM();
    int* s = seq;
    *g_reord = bitmap_reorder(*g_mat, sect, s);
    std::memcpy(indptr, g_reord->indptr, sizeof(int) * (g_reord->shape[0] + 1));
    std::memcpy(indices, g_reord->indices, sizeof(int) * g_reord->shape[2]);
    std::memcpy(data, g_reord->data, sizeof(double) * g_reord->shape[2]);
    return 0;
}

// gen_new_panels 4-arg (transmat.h:334) on the reordered matrix.
int ref_gen_new_panels(int* bsize_list, int* bnum) {
    if (!g_reord) return -1;
    bsize_list[0] = 0;
    int end = 1, bn = 0;
    int* p = bsize_list;
    gen_new_panels(*g_reord, p, end, bn);
    *bnum = bn;
    return end;
}

// One block [row_beg,row_end) of the reordered matrix through the reference's
// per-block functions.  Buffers: panel_list[rows/8+2], out_seq[rows],
// spv8_list[panels], b_indptr[rows+1], b_colidx[nnz_b] (interleaved original ids),
// b_data[nnz_b], b_local[nnz_b], keys[nnz_b].
// Returns the number of distinct columns; *n_panel_list receives the list length.
int ref_stage_block(int row_beg, int row_end, int* panel_list, int* n_panel_list, int* out_seq,
                    int* spv8_out, int* b_indptr, int* b_colidx, double* b_data, int* b_local,
                    int* keys) {
    if (!g_reord) return -1;
    int npl = 0;
    int* pl = panel_list;
    gen_panel_list(*g_reord, row_beg, row_end, pl, npl);
    *n_panel_list = npl;
    int end_out = 0;
    int* os = out_seq;
    std::vector<int> spv8;
    panel_sort_nnz(os, end_out, spv8, *g_reord, row_beg, row_end, pl, npl);
    for (size_t i = 0; i < spv8.size(); i++) spv8_out[i] = spv8[i];
    b_indptr[0] = 0;
    SpM tmp = reorder_row(*g_reord, row_beg, row_end, out_seq, b_indptr);
    transpose_spv8_nnz(tmp, spv8, pl, b_colidx, b_data);
    int nk = 0;
    int* kp = keys;
    gen_serial_origin_vec(tmp.shape, b_colidx, b_local, kp, nk);
    return nk;
}

// SerialSort_block_vec + gen_rseq + SeqReverse (wbsort.h:81,58,16) with the
// per-block orders handed in as one concatenated array.
int ref_wbseq(int nrow, int* seq_bitmap, int* out_seq_all, int* bsize_list, int bnum, int* keys,
              int nkeys, int* seq_input, int* rseq) {
    std::vector<int*> blocks(bnum);
    std::vector<int> lens(bnum);
    int seq_len = 0;
    for (int b = 0; b < bnum; b++) {
        blocks[b] = out_seq_all + bsize_list[b];
        lens[b] = bsize_list[b + 1] - bsize_list[b];
        seq_len += lens[b];
    }
    int end_in = 0;
    int* si = seq_input;
    SerialSort_block_vec(seq_bitmap, blocks.data(), lens.data(), keys, nkeys, si, end_in,
                         bsize_list, bnum, seq_len);
    int* seq_row = gen_rseq(seq_bitmap, blocks.data(), lens.data(), bsize_list, bnum, seq_len);
    int* r = SeqReverse(seq_row, seq_len);
    std::memcpy(rseq, r, sizeof(int) * seq_len);
    delete[] seq_row;
    delete[] r;
    (void)nrow;
    return end_in;
}

}  // extern "C"
