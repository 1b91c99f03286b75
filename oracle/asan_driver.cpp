// Sanitizer run of the CPU oracle.  TEST INFRASTRUCTURE ONLY (like everything under oracle/).
//
// Built by `make -C oracle asan` together with spmm_oracle.cpp under -fsanitize=address,undefined
// into oracle/_asan/oracle_asan.  It reads one matrix in coordinate form from a binary file
// (int32 nrow, ncol, nnz, then nnz row ids and nnz column ids), runs every stage of the oracle on
// heap buffers of exactly the sizes the callers' contract names, and prints one line
// "<array> <FNV-1a-64 of its 32-bit words>" per output.  tests/test_oracle_asan.py compares the
// lines with what the ordinary liboracle.so gives on the same input: the sanitizers abort on the
// first out-of-bounds access, use of uninitialised heap size, signed overflow or misaligned access,
// which is how the overflow of the reference's panel sort (v8sort.h:157-160: scratch sized by
// block nnz, indexed by block rows) was found.
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
struct orc_trace_out {
    int32_t* seq_bitmap; int32_t* r_indptr; int32_t* r_indices; double* r_data;
    int32_t* bsize_list; int32_t* seq_offset; int32_t cap_blocks;
    int32_t* panel_ptr; int32_t* panel_list; int32_t* spv8_list;
    int32_t* out_seq_all; int32_t* b_indptr; int32_t* b_colidx_orig; int32_t* b_local;
    double* b_data; int32_t* keys; int32_t* seq_input; int32_t* rseq;
    int32_t bnum; int32_t n_keys; int32_t n_panel_entries; int32_t n_spv8;
};
void orc_csr_from_coo(int32_t nrow, int64_t nnz, const int32_t* row, const int32_t* col, int32_t* indptr, int32_t* indices,
                      double* data);
int32_t orc_trace_formats(int32_t nrow, int32_t ncol, const int32_t* indptr, const int32_t* indices, const double* data,
                          int32_t sect, int32_t threshold, int sort_mode, orc_trace_out* o);
int64_t orc_spgemm_symbolic(int32_t nrow, int32_t ncolB, const int32_t* a_indptr, const int32_t* a_indices,
                            const int32_t* b_indptr, const int32_t* b_indices, int64_t* c_indptr);
int64_t orc_spgemm_numeric(int32_t row_beg, int32_t row_end, int32_t ncolB, const int32_t* a_indptr, const int32_t* a_indices,
                           const double* a_data, const int32_t* b_indptr, const int32_t* b_indices, const double* b_data,
                           const int64_t* c_indptr, int32_t* c_indices, double* c_data);
int64_t orc_spgemm_rows_omp(int32_t nlist, const int32_t* rows, int32_t ncolB, const int32_t* a_indptr, const int32_t* a_indices,
                            const double* a_data, const int32_t* b_indptr, const int32_t* b_indices, const double* b_data,
                            int32_t threads, int64_t* counts, double* value_sum);
int64_t orc_row_products(int32_t nrow, const int32_t* a_indptr, const int32_t* a_indices, const int32_t* b_indptr, int64_t* prod);
uint64_t orc_fnv1a64_u32(const uint32_t* v, int64_t n);
}

namespace {
template <typename T>
void line(const char* name, const T* v, size_t n) {
    std::printf("%s %016" PRIx64 "\n", name, orc_fnv1a64_u32((const uint32_t*)v, (int64_t)(n * sizeof(T) / 4)));
}
// heap arrays of exactly n elements (at least one byte, so that a zero-sized one is still a distinct allocation)
template <typename T>
struct arr {
    T* p;
    size_t n;
    explicit arr(size_t n_) : p((T*)std::malloc(n_ * sizeof(T) + (n_ == 0))), n(n_) {}
    ~arr() { std::free(p); }
    arr(const arr&) = delete;
};
}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: oracle_asan <input.bin> [threshold]\n"); return 2; }
    const int threshold = argc > 2 ? std::atoi(argv[2]) : 65536;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    int32_t hdr[3];
    if (std::fread(hdr, 4, 3, f) != 3) return 2;
    const int32_t nrow = hdr[0], ncol = hdr[1], nnz = hdr[2];
    arr<int32_t> row(nnz), col(nnz);
    if (std::fread(row.p, 4, nnz, f) != (size_t)nnz || std::fread(col.p, 4, nnz, f) != (size_t)nnz) return 2;
    std::fclose(f);

    arr<int32_t> ip(nrow + 1), ix(nnz);
    arr<double> d(nnz);
    orc_csr_from_coo(nrow, nnz, row.p, col.p, ip.p, ix.p, d.p);
    line("csr_indptr", ip.p, ip.n);
    line("csr_indices", ix.p, ix.n);

    if (ncol <= nrow)   // the write-back sequence indexes rseq by column id (wbsort.h:88)
        for (int mode = 0; mode < 2; mode++) {
            const size_t cap = (size_t)nrow + 1, npl = (size_t)nrow / 8 + 3 * cap + 8;
            arr<int32_t> seq_bitmap(nrow), r_indptr(nrow + 1), r_indices(nnz), bsize(cap + 1), seq_offset(cap + 1), panel_ptr(cap + 1),
                panel_list(npl), spv8_list(npl), out_seq(nrow), b_indptr(nrow + cap + 1), b_col(nnz), b_local(nnz), keys(nnz),
                seq_input(nnz), rseq(nrow);
            arr<double> r_data(nnz), b_data(nnz);
            orc_trace_out o{seq_bitmap.p, r_indptr.p, r_indices.p, r_data.p, bsize.p, seq_offset.p, (int32_t)cap, panel_ptr.p,
                            panel_list.p, spv8_list.p, out_seq.p, b_indptr.p, b_col.p, b_local.p, b_data.p, keys.p, seq_input.p,
                            rseq.p, 0, 0, 0, 0};
            const int32_t bnum = orc_trace_formats(nrow, ncol, ip.p, ix.p, d.p, 2048, threshold, mode, &o);
            if (bnum < 0) { std::fprintf(stderr, "orc_trace_formats failed\n"); return 1; }
            char nm[64];
            auto out = [&](const char* what, const int32_t* v, size_t n) {
                std::snprintf(nm, sizeof nm, "%s%s", mode ? "stable_" : "ref_", what);
                line(nm, v, n);
            };
            std::printf("%sbnum %d\n", mode ? "stable_" : "ref_", bnum);
            out("seq_bitmap", seq_bitmap.p, nrow);
            out("r_indptr", r_indptr.p, (size_t)nrow + 1);
            out("r_indices", r_indices.p, nnz);
            out("bsize_list", bsize.p, (size_t)bnum + 1);
            out("seq_offset", seq_offset.p, (size_t)bnum + 1);
            out("panel_list", panel_list.p, o.n_panel_entries);
            out("spv8_list", spv8_list.p, o.n_spv8);
            out("out_seq", out_seq.p, nrow);
            out("b_indptr", b_indptr.p, (size_t)nrow + bnum);
            out("b_colidx", b_col.p, nnz);
            out("b_local", b_local.p, nnz);
            out("keys", keys.p, o.n_keys);
            out("seq_input", seq_input.p, o.n_keys);
            out("rseq", rseq.p, nrow);
        }

    if (nrow == ncol) {   // C = A*A
        arr<int64_t> c_ip((size_t)nrow + 1), prod(nrow);
        const int64_t nnzc = orc_spgemm_symbolic(nrow, ncol, ip.p, ix.p, ip.p, ix.p, c_ip.p);
        const int64_t P = orc_row_products(nrow, ip.p, ix.p, ip.p, prod.p);
        arr<int32_t> c_ix(nnzc);
        arr<double> c_d(nnzc);
        // in two row ranges, each packed from position 0 of its own buffer
        const int32_t mid = nrow / 2;
        const int64_t n0 = c_ip.p[mid], n1 = nnzc - n0;
        arr<int32_t> c_ix1(n1);
        arr<double> c_d1(n1);
        int64_t P2 = orc_spgemm_numeric(0, mid, ncol, ip.p, ix.p, d.p, ip.p, ix.p, d.p, c_ip.p, c_ix.p, c_d.p);
        P2 += orc_spgemm_numeric(mid, nrow, ncol, ip.p, ix.p, d.p, ip.p, ix.p, d.p, c_ip.p, c_ix1.p, c_d1.p);
        if (P2 != P) { std::fprintf(stderr, "products: %lld from the numeric calls, %lld from the row counts\n", (long long)P2, (long long)P); return 1; }
        for (int64_t i = 0; i < n1; i++) { c_ix.p[n0 + i] = c_ix1.p[i]; c_d.p[n0 + i] = c_d1.p[i]; }
        line("c_indptr", c_ip.p, c_ip.n);
        line("c_indices", c_ix.p, c_ix.n);
        line("c_data", c_d.p, c_d.n);
        line("row_products", prod.p, prod.n);
        // the row-list form on every third row, two threads
        std::vector<int32_t> rows;
        for (int32_t r = 0; r < nrow; r += 3) rows.push_back(r);
        arr<int64_t> counts(rows.size());
        double vs = 0.0;
        orc_spgemm_rows_omp((int32_t)rows.size(), rows.data(), ncol, ip.p, ix.p, d.p, ip.p, ix.p, d.p, 2, counts.p, &vs);
        line("row_list_counts", counts.p, counts.n);
    }
    return 0;
}
