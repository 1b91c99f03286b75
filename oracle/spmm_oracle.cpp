// CPU oracle for the pre-processing pipeline and the sparse product C = A*B.
//
// TEST INFRASTRUCTURE ONLY.  This file is a serial restatement of the algorithms
// in the reference tree (PreProcessing/*.h, serial_newblock_clock.cpp).  It is
// used by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg
// as the checker / reported baseline.  Nothing in the shipped library calls it.
//
// Parity status
//   * stages orc_csr_from_coo .. orc_wbseq and orc_trace_formats: pinned against
//     the compiled reference (oracle/_ref/libref.so built from the reference's
//     own sources by oracle/Makefile) through the fixtures in tests/golden/.
//   * orc_spgemm_*: the reference contains no sparse product at all
//     (serial_newblock_clock.cpp:501-599 only times the pre-processing), so the
//     product is PARITY UNPINNED by the reference.  It is pinned instead by
//     golden vectors generated with scipy.sparse (tools/gen_fixtures.py).
//
// Every function cites the reference lines it follows.  Allocation defects of
// the reference (v8sort.h:157 scratch sized by nnz instead of rows, fixed
// 1000-entry block lists at serial_newblock_clock.cpp:552-553, the
// (ncol/2048+1) x nrow pool at bitmap.h:119-121) are NOT reproduced: outputs are
// identical wherever the reference survives.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

extern "C" {

// ---------------------------------------------------------------------------
// MatrixMarket coordinate reader: serial_newblock_clock.cpp:47-104.
// '%' lines skipped, size line "nrow ncol nnz", then nnz lines "row col [value]"
// (1-based).  The value token, if any, is ignored: every stored value is 1.0
// (serial_newblock_clock.cpp:84,96).  Pass 1 (count_only=1) returns the sizes.
// Returns 0 on success, <0 on error (the reference never checks; it reads garbage).
int orc_read_mtx(const char* path, int32_t* nrow, int32_t* ncol, int64_t* nnz,
                 int32_t* row, int32_t* col, int count_only) {
    FILE* f = std::fopen(path, "r");
    if (!f) return -1;
    char line[4096];
    bool have_size = false;
    long long nr = 0, nc = 0, nz = 0;
    while (std::fgets(line, sizeof line, f)) {
        if (line[0] == '%') continue;
        if (std::sscanf(line, "%lld %lld %lld", &nr, &nc, &nz) == 3) { have_size = true; }
        break;
    }
    if (!have_size) { std::fclose(f); return -2; }
    *nrow = (int32_t)nr; *ncol = (int32_t)nc; *nnz = nz;
    if (count_only) { std::fclose(f); return 0; }
    for (long long l = 0; l < nz; l++) {
        if (!std::fgets(line, sizeof line, f)) { std::fclose(f); return -3; }
        long long r, c;
        if (std::sscanf(line, "%lld %lld", &r, &c) != 2) { std::fclose(f); return -4; }
        row[l] = (int32_t)(r - 1);
        col[l] = (int32_t)(c - 1);
    }
    std::fclose(f);
    return 0;
}

// ---------------------------------------------------------------------------
// COO -> CSR, stable in input order: serial_newblock_clock.cpp:78-112.
// Row histogram (:87), exclusive scan (:105-107), scatter that keeps the file
// order inside a row (:108-112).  Values are all 1.0 (:84).  Duplicates kept.
void orc_csr_from_coo(int32_t nrow, int64_t nnz, const int32_t* row, const int32_t* col,
                      int32_t* indptr, int32_t* indices, double* data) {
    std::vector<int32_t> fill((size_t)nrow, 0);
    for (int64_t l = 0; l < nnz; l++) fill[row[l]]++;
    indptr[0] = 0;
    for (int32_t i = 0; i < nrow; i++) { indptr[i + 1] = indptr[i] + fill[i]; fill[i] = 0; }
    for (int64_t l = 0; l < nnz; l++) {
        int32_t r = row[l];
        indices[indptr[r] + fill[r]++] = col[l];
    }
    for (int64_t l = 0; l < nnz; l++) data[l] = 1.0;
}

// ---------------------------------------------------------------------------
// Row key of the first reorder: bitmap.h:123-143.
// Walk the row's columns in stored order, sec = col / sect.  When the section
// index rises above the previous one and the running score beats the best so
// far, the PREVIOUS section becomes the best and the score restarts; the score
// is not reset otherwise, the last section is never compared, '>' is strict.
// key = best section + 1 (0 when no section ever won).
static inline int32_t row_key(const int32_t* cols, int32_t len, int32_t sect) {
    int32_t best = 0, best_sec = -1, prev_sec = 0, score = 0;
    for (int32_t j = 0; j < len; j++) {
        int32_t sec = cols[j] / sect;
        if (sec > prev_sec && score > best) { best = score; best_sec = prev_sec; score = 0; }
        prev_sec = sec;
        score++;
    }
    return best_sec + 1;
}

void orc_row_keys(int32_t nrow, const int32_t* indptr, const int32_t* indices, int32_t sect,
                  int32_t* keys) {
    for (int32_t i = 0; i < nrow; i++)
        keys[i] = row_key(indices + indptr[i], indptr[i + 1] - indptr[i], sect);
}

// Row gather: out row t = in row perm[t].  bitmap.h:147-163 (and the twin
// transmat.h:135-169).
static void gather_rows(int32_t nrow_out, const int32_t* perm, const int32_t* indptr,
                        const int32_t* indices, const double* data, int32_t* out_indptr,
                        int32_t* out_indices, double* out_data) {
    out_indptr[0] = 0;
    int32_t tail = 0;
    for (int32_t t = 0; t < nrow_out; t++) {
        int32_t s = perm[t];
        for (int32_t k = indptr[s]; k < indptr[s + 1]; k++) {
            out_indices[tail] = indices[k];
            out_data[tail] = data[k];
            tail++;
        }
        out_indptr[t + 1] = tail;
    }
}

// First reorder: bitmap.h:108-170.  Stable bucket of the rows by key
// (keys in [0, ncol/sect], bitmap.h:110 computes ceil of an already-integer
// quotient), seq[new] = old, then the permuted CSR.
void orc_bitmap_reorder(int32_t nrow, int32_t ncol, const int32_t* indptr, const int32_t* indices,
                        const double* data, int32_t sect, int32_t* out_indptr,
                        int32_t* out_indices, double* out_data, int32_t* seq) {
    int32_t nkeys = ncol / sect + 1;
    std::vector<int32_t> keys((size_t)nrow);
    orc_row_keys(nrow, indptr, indices, sect, keys.data());
    std::vector<int32_t> start((size_t)nkeys + 1, 0);
    for (int32_t i = 0; i < nrow; i++) start[keys[i] + 1]++;
    for (int32_t k = 0; k < nkeys; k++) start[k + 1] += start[k];
    for (int32_t i = 0; i < nrow; i++) seq[start[keys[i]]++] = i;
    gather_rows(nrow, seq, indptr, indices, data, out_indptr, out_indices, out_data);
}

// ---------------------------------------------------------------------------
// Greedy row blocking by distinct column count: transmat.h:334-376.
// Rows are appended to the current block; once the number of distinct columns
// seen in the block reaches `threshold` (65536 in the reference, :339) the block
// is cut AFTER that row.  nrow is always appended (:366), so a cut on the last
// row leaves an empty final block.  bsize_list[0] = 0 is the caller's
// (serial_newblock_clock.cpp:264).  Returns bnum = number of blocks; at most
// `cap` entries are written, -1 is returned when cap is too small (the
// reference overflows its 1000-int buffer instead).
int32_t orc_gen_blocks(int32_t nrow, int32_t ncol, const int32_t* indptr, const int32_t* indices,
                       int32_t threshold, int32_t* bsize_list, int32_t cap) {
    std::vector<uint8_t> seen((size_t)ncol, 0);
    std::vector<int32_t> touched;
    int32_t end = 0, counter = 0;
    if (cap < 2) return -1;
    bsize_list[end++] = 0;
    for (int32_t r = 0; r < nrow; r++) {
        for (int32_t k = indptr[r]; k < indptr[r + 1]; k++) {
            int32_t c = indices[k];
            if (!seen[c]) { seen[c] = 1; counter++; touched.push_back(c); }
        }
        if (counter >= threshold) {
            for (int32_t c : touched) seen[c] = 0;
            touched.clear();
            counter = 0;
            if (end >= cap) return -1;
            bsize_list[end++] = r + 1;
        }
    }
    if (end >= cap) return -1;
    bsize_list[end++] = nrow;
    return end - 1;
}

// ---------------------------------------------------------------------------
// Panel boundaries inside one block: v8sort.h:49-73.
// panel_num = rows/2048 + 1, panel_size = nnz/panel_num + 1; rows are visited in
// steps of 8 and a boundary is placed at row i when the block-local nnz offset
// of row i exceeds the running bound.  Output is block-local: [0, ..., rows].
// Returns the list length.
int32_t orc_panel_list(const int32_t* indptr, int32_t row_beg, int32_t row_end,
                       int32_t* panel_list) {
    int32_t rows = row_end - row_beg;
    const int32_t* ip = indptr + row_beg;
    int32_t off = ip[0];
    int32_t nnz = ip[rows] - off;
    int32_t panel_num = rows / 2048 + 1;
    int32_t panel_size = nnz / panel_num + 1;
    int64_t bound = panel_size;
    int32_t n = 0;
    panel_list[n++] = 0;
    for (int32_t i = 0; i < rows; i += 8)
        if (ip[i] - off > bound) { bound += panel_size; panel_list[n++] = i; }
    panel_list[n++] = rows;
    return n;
}

// ---------------------------------------------------------------------------
// Per-panel row order: v8sort.h:152-232 (+ bitmap.h:47-52, v8sort.h:19-25).
// Rows of each panel are argsorted by length.  For every length L in 0..32 the
// first cnt(L) - cnt(L)%8 rows of that length go to the "grouped" part, the
// leftovers and every row longer than 32 go to the "remain" part in sorted
// order; the panel's output is grouped ++ remain and spv8_list[panel] is the
// size of the grouped part.  Ids are block-local.
//   mode 0: std::sort with the reference's comparator (bitmap.h:50) -- ties in
//           libstdc++'s introsort order, bit-identical to the reference binary.
//   mode 1: stable by (length, block-local row id) -- the canonical order the
//           GPU implementation produces.
void orc_panel_sort(const int32_t* indptr, int32_t row_beg, int32_t row_end,
                    const int32_t* panel_list, int32_t n_panel_list, int mode, int32_t* out_seq,
                    int32_t* spv8_list) {
    int32_t rows = row_end - row_beg;
    const int32_t* ip = indptr + row_beg;
    std::vector<int32_t> len((size_t)rows);
    for (int32_t r = 0; r < rows; r++) len[r] = ip[r + 1] - ip[r];
    std::vector<int32_t> idx, grouped, remain;
    int32_t tail = 0;
    for (int32_t j = 0; j + 1 < n_panel_list; j++) {
        int32_t p0 = panel_list[j], pn = panel_list[j + 1] - panel_list[j];
        const int32_t* plen = len.data() + p0;
        idx.resize(pn);
        std::iota(idx.begin(), idx.end(), 0);
        auto by_len = [plen](int32_t a, int32_t b) { return plen[a] < plen[b]; };
        if (mode == 0) std::sort(idx.begin(), idx.end(), by_len);
        else std::stable_sort(idx.begin(), idx.end(), by_len);
        int32_t cnt[33] = {0};
        for (int32_t i = 0; i < pn; i++) if (plen[i] < 33) cnt[plen[i]]++;
        grouped.clear(); remain.clear();
        int32_t pos = 0;
        for (int32_t L = 0; L < 33; L++) {
            int32_t full = cnt[L] - cnt[L] % 8;
            for (int32_t i = 0; i < full; i++) grouped.push_back(idx[pos + i] + p0);
            for (int32_t i = full; i < cnt[L]; i++) remain.push_back(idx[pos + i] + p0);
            pos += cnt[L];
        }
        for (int32_t i = pos; i < pn; i++) remain.push_back(idx[i] + p0);
        spv8_list[j] = (int32_t)grouped.size();
        for (int32_t v : grouped) out_seq[tail++] = v;
        for (int32_t v : remain) out_seq[tail++] = v;
    }
}

// ---------------------------------------------------------------------------
// Block row gather + 8-row interleave: serial_newblock_clock.cpp:336-399
// (readable twins transmat.h:135-169 and v8sort.h:300-393).
// Rows of the block are gathered in out_seq order (block-local indptr starting
// at 0).  Inside each panel the first spv8_list[panel] rows form groups of 8
// equal-length rows; a group of length L is rewritten in place so that
// out[8*r + c] = in[c*L + r] (r < L, c < 8).  Remaining rows are left as they are.
// The reference reads the VALUES of a group from the un-gathered block
// (serial_newblock_clock.cpp:376); all values are 1.0 there, so gathering them
// like the column ids, as done here, gives the same bytes.
void orc_gather_spv8(const int32_t* indptr, const int32_t* indices, const double* data,
                     int32_t row_beg, int32_t row_end, const int32_t* out_seq,
                     const int32_t* panel_list, int32_t n_panel_list, const int32_t* spv8_list,
                     int32_t* b_indptr, int32_t* b_colidx, double* b_data) {
    int32_t rows = row_end - row_beg;
    gather_rows(rows, out_seq, indptr + row_beg, indices, data, b_indptr, b_colidx, b_data);
    // gather_rows produced absolute-source copies; b_indptr is already block-local.
    std::vector<int32_t> tc;
    std::vector<double> td;
    for (int32_t j = 0; j + 1 < n_panel_list; j++) {
        int32_t base = panel_list[j];
        for (int32_t g = base; g < base + spv8_list[j]; g += 8) {
            int32_t L = b_indptr[g + 1] - b_indptr[g];
            int32_t at = b_indptr[g];
            tc.assign(b_colidx + at, b_colidx + at + 8 * (size_t)L);
            td.assign(b_data + at, b_data + at + 8 * (size_t)L);
            for (int32_t r = 0; r < L; r++)
                for (int32_t c = 0; c < 8; c++) {
                    b_colidx[at + 8 * r + c] = tc[(size_t)c * L + r];
                    b_data[at + 8 * r + c] = td[(size_t)c * L + r];
                }
        }
    }
}

// ---------------------------------------------------------------------------
// First-touch column renumbering over one block's column stream:
// serial_newblock_clock.cpp:187-204.  The first occurrence of a column gets
// the next local id and its original id is appended to keys; later occurrences
// reuse the id.  Returns the number of distinct columns.
int32_t orc_first_touch(int32_t ncol, int64_t n, const int32_t* colidx, int32_t* local_ids,
                        int32_t* keys) {
    std::vector<int32_t> id((size_t)ncol, -1);
    int32_t next = 0;
    for (int64_t i = 0; i < n; i++) {
        int32_t c = colidx[i];
        if (id[c] < 0) { id[c] = next; keys[next] = c; next++; }
        local_ids[i] = id[c];
    }
    return next;
}

// ---------------------------------------------------------------------------
// Write-back / fetch sequence: wbsort.h:58-67 (compose), :16-34 (invert),
// :81-95 (gather); serial_newblock_clock.cpp:471-484.
//   seq_row[t]  = seq_bitmap[out_seq_all[t] + bsize[block of t]]   (new -> original row)
//   rseq[seq_row[t]] = t                                          (original -> new)
//   seq_input[i] = rseq[keys[i]]
// out_seq_all is the concatenation of the per-block orders (block-local ids).
void orc_wbseq(int32_t nrow, const int32_t* seq_bitmap, const int32_t* out_seq_all,
               const int32_t* bsize_list, int32_t bnum, const int32_t* keys, int64_t nkeys,
               int32_t* seq_input, int32_t* rseq) {
    int32_t t = 0;
    for (int32_t b = 0; b < bnum; b++)
        for (int32_t r = bsize_list[b]; r < bsize_list[b + 1]; r++, t++)
            rseq[seq_bitmap[out_seq_all[t] + bsize_list[b]]] = t;
    (void)nrow;
    for (int64_t i = 0; i < nkeys; i++) seq_input[i] = rseq[keys[i]];
}

// ---------------------------------------------------------------------------
// Whole pipeline: serial_newblock_clock.cpp:221-499.  All outputs are
// caller-allocated at their worst-case sizes:
//   seq_bitmap[nrow], r_indptr[nrow+1], r_indices[nnz], r_data[nnz]  (after first reorder)
//   bsize_list[cap_blocks+1], seq_offset[cap_blocks+1]
//   panel_ptr[cap_blocks+1] (offsets into panel_list / spv8 arrays), panel_list[nrow/8+2*cap+..]
//   out_seq_all[nrow], spv8_list[...]
//   b_indptr[nrow + bnum]  (per block rows+1 entries, block-local, concatenated)
//   b_colidx_orig[nnz] (interleaved, ORIGINAL column ids), b_local[nnz] (first-touch ids), b_data[nnz]
//   keys[nnz], seq_input[nnz], rseq[nrow]
// Returns bnum or <0.
struct orc_trace_out {
    int32_t* seq_bitmap; int32_t* r_indptr; int32_t* r_indices; double* r_data;
    int32_t* bsize_list; int32_t* seq_offset; int32_t cap_blocks;
    int32_t* panel_ptr; int32_t* panel_list; int32_t* spv8_list;
    int32_t* out_seq_all; int32_t* b_indptr; int32_t* b_colidx_orig; int32_t* b_local;
    double* b_data; int32_t* keys; int32_t* seq_input; int32_t* rseq;
    int32_t bnum; int32_t n_keys; int32_t n_panel_entries; int32_t n_spv8;
};

int32_t orc_trace_formats(int32_t nrow, int32_t ncol, const int32_t* indptr,
                          const int32_t* indices, const double* data, int32_t sect,
                          int32_t threshold, int sort_mode, orc_trace_out* o) {
    orc_bitmap_reorder(nrow, ncol, indptr, indices, data, sect, o->r_indptr, o->r_indices,
                       o->r_data, o->seq_bitmap);
    int32_t bnum = orc_gen_blocks(nrow, ncol, o->r_indptr, o->r_indices, threshold,
                                  o->bsize_list, o->cap_blocks + 1);
    if (bnum < 0) return -1;
    o->bnum = bnum;
    o->seq_offset[0] = 0;
    int32_t pl_tail = 0, sp_tail = 0, nkeys = 0, ip_tail = 0;
    for (int32_t b = 0; b < bnum; b++) {
        int32_t r0 = o->bsize_list[b], r1 = o->bsize_list[b + 1];
        int32_t nz0 = o->r_indptr[r0];
        o->panel_ptr[b] = pl_tail;
        int32_t npl = orc_panel_list(o->r_indptr, r0, r1, o->panel_list + pl_tail);
        orc_panel_sort(o->r_indptr, r0, r1, o->panel_list + pl_tail, npl, sort_mode,
                       o->out_seq_all + r0, o->spv8_list + sp_tail);
        orc_gather_spv8(o->r_indptr, o->r_indices, o->r_data, r0, r1, o->out_seq_all + r0,
                        o->panel_list + pl_tail, npl, o->spv8_list + sp_tail,
                        o->b_indptr + ip_tail, o->b_colidx_orig + nz0, o->b_data + nz0);
        int32_t bn = o->r_indptr[r1] - nz0;
        nkeys += orc_first_touch(ncol, bn, o->b_colidx_orig + nz0, o->b_local + nz0,
                                 o->keys + nkeys);
        o->seq_offset[b + 1] = nkeys;
        pl_tail += npl;
        sp_tail += npl - 1;
        ip_tail += (r1 - r0) + 1;
    }
    o->panel_ptr[bnum] = pl_tail;
    o->n_panel_entries = pl_tail;
    o->n_spv8 = sp_tail;
    o->n_keys = nkeys;
    orc_wbseq(nrow, o->seq_bitmap, o->out_seq_all, o->bsize_list, bnum, o->keys, nkeys,
              o->seq_input, o->rseq);
    return bnum;
}

// ---------------------------------------------------------------------------
// Sparse product C = A*B, row-wise Gustavson with a dense accumulator.
// NOT IN THE REFERENCE (parity unpinned by it; pinned by scipy goldens).
// The product the pipeline prepares for is implied by wbsort.h:87-88: a column
// id of A is mapped through A's row permutation, i.e. column k selects row k of
// the same matrix, so B = A.  Output rows have ascending, distinct column ids;
// row pointers are 64-bit.  Symbolic returns nnz(C); c_indptr has nrow+1 entries.
int64_t orc_spgemm_symbolic(int32_t nrow, int32_t ncolB, const int32_t* a_indptr,
                            const int32_t* a_indices, const int32_t* b_indptr,
                            const int32_t* b_indices, int64_t* c_indptr) {
    std::vector<int32_t> mark((size_t)ncolB, -1);
    c_indptr[0] = 0;
    for (int32_t i = 0; i < nrow; i++) {
        int64_t cnt = 0;
        for (int32_t p = a_indptr[i]; p < a_indptr[i + 1]; p++) {
            int32_t k = a_indices[p];
            for (int32_t q = b_indptr[k]; q < b_indptr[k + 1]; q++) {
                int32_t c = b_indices[q];
                if (mark[c] != i) { mark[c] = i; cnt++; }
            }
        }
        c_indptr[i + 1] = c_indptr[i] + cnt;
    }
    return c_indptr[nrow];
}

// Numeric phase for rows [row_beg, row_end).  c_indices / c_data receive the
// rows packed from position 0 (i.e. offset c_indptr[row_beg] is subtracted).
// Products are summed in the order (A entry, B entry); values that are small
// integers, as for the reference's all-ones matrices, are therefore exact.
// Returns the number of intermediate products of the row range.
int64_t orc_spgemm_numeric(int32_t row_beg, int32_t row_end, int32_t ncolB,
                           const int32_t* a_indptr, const int32_t* a_indices, const double* a_data,
                           const int32_t* b_indptr, const int32_t* b_indices, const double* b_data,
                           const int64_t* c_indptr, int32_t* c_indices, double* c_data) {
    std::vector<int32_t> mark((size_t)ncolB, -1);
    std::vector<double> acc((size_t)ncolB, 0.0);
    std::vector<int32_t> cols;
    int64_t base = c_indptr[row_beg], products = 0;
    for (int32_t i = row_beg; i < row_end; i++) {
        cols.clear();
        for (int32_t p = a_indptr[i]; p < a_indptr[i + 1]; p++) {
            int32_t k = a_indices[p];
            double av = a_data[p];
            for (int32_t q = b_indptr[k]; q < b_indptr[k + 1]; q++) {
                int32_t c = b_indices[q];
                if (mark[c] != i) { mark[c] = i; acc[c] = 0.0; cols.push_back(c); }
                acc[c] += av * b_data[q];
                products++;
            }
        }
        std::sort(cols.begin(), cols.end());
        int64_t at = c_indptr[i] - base;
        for (size_t t = 0; t < cols.size(); t++) {
            c_indices[at + (int64_t)t] = cols[t];
            c_data[at + (int64_t)t] = acc[cols[t]];
        }
    }
    return products;
}

// The same Gustavson product over a LIST of rows, row-parallel over `threads` OpenMP threads (1 = serial): the CPU
// baseline timed beside the GPU.  Every thread keeps its own dense accumulator; a row's sorted column ids and sums
// are produced and folded into counts[t] (entries of C of rows[t]) and *value_sum (sum of all values), not stored.
// NOT IN THE REFERENCE, like the product itself.  Returns the intermediate products of the listed rows.
int64_t orc_spgemm_rows_omp(int32_t nlist, const int32_t* rows, int32_t ncolB, const int32_t* a_indptr,
                            const int32_t* a_indices, const double* a_data, const int32_t* b_indptr,
                            const int32_t* b_indices, const double* b_data, int32_t threads, int64_t* counts,
                            double* value_sum) {
    int64_t products = 0;
    double vsum = 0.0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads) reduction(+ : products, vsum)
    {
        std::vector<int32_t> mark((size_t)ncolB, -1);
        std::vector<double> acc((size_t)ncolB, 0.0);
        std::vector<int32_t> cols;
#pragma omp for schedule(dynamic, 1)
        for (int32_t t = 0; t < nlist; t++) {
            const int32_t i = rows[t];
            cols.clear();
            for (int32_t p = a_indptr[i]; p < a_indptr[i + 1]; p++) {
                const int32_t k = a_indices[p];
                const double av = a_data[p];
                for (int32_t q = b_indptr[k]; q < b_indptr[k + 1]; q++) {
                    const int32_t c = b_indices[q];
                    if (mark[c] != t) { mark[c] = t; acc[c] = 0.0; cols.push_back(c); }
                    acc[c] += av * b_data[q];
                    products++;
                }
            }
            std::sort(cols.begin(), cols.end());
            double s = 0.0;
            for (size_t u = 0; u < cols.size(); u++) s += acc[cols[u]];
            vsum += s;
            counts[t] = (int64_t)cols.size();
        }
    }
    if (value_sum) *value_sum = vsum;
    return products;
}

// Intermediate products per row: prod[i] = sum over A entries (i,k) of len_B(k).
int64_t orc_row_products(int32_t nrow, const int32_t* a_indptr, const int32_t* a_indices,
                         const int32_t* b_indptr, int64_t* prod) {
    int64_t total = 0;
    for (int32_t i = 0; i < nrow; i++) {
        int64_t s = 0;
        for (int32_t p = a_indptr[i]; p < a_indptr[i + 1]; p++) {
            int32_t k = a_indices[p];
            s += b_indptr[k + 1] - b_indptr[k];
        }
        if (prod) prod[i] = s;
        total += s;
    }
    return total;
}

// FNV-1a-64 over 32-bit words (one xor+multiply per word); used for golden digests.
uint64_t orc_fnv1a64_u32(const uint32_t* v, int64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (int64_t i = 0; i < n; i++) { h ^= (uint64_t)v[i]; h *= 1099511628211ull; }
    return h;
}

}  // extern "C"
