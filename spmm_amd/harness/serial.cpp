// Drop-in replacement of the reference's ./serial (serial_newblock_clock.cpp:501-599):
//   matrix.txt (one name per line; text from the first '.' is dropped, :36-45)
//   -> mat/mtx/<name>/<name>.mtx (:535-538)
//   -> result.txt, truncated at start, one line "<name> <ms>ms" per matrix (:510-512, :565)
//   -> stdout "----name:<name>----" (:567); exit 1 with a message when result.txt cannot be
//      opened (:513-516).
// The work is done by libspmm_hip.so through its C ABI: spmm_mtx_header / spmm_mtx_entries in place of
// the reading part of csr_matrix() (:47-104), spmm_csr_from_coo in place of its tail (:78-112) and spmm_trace_formats in place of gen_trace_formats()
// (:221-499).  There is no CPU path: without a gfx950 GPU the program stops with an error.
//
// The time written to result.txt is the wall time of spmm_trace_formats from the CSR in host memory to the
// outputs back in host memory (PCIe both ways included): the reference's figure (:565) is the sum of its stage
// timers and likewise starts with the matrix resident in host memory.  SPMM_HARNESS_TIME=gpu writes the GPU time
// of the stages instead (hipEvents, matrix resident in device memory); result_spgemm.txt carries both.  With SPMM_HARNESS_SPGEMM=1 the
// product C = A*A is computed as well and one JSON line per matrix goes to result_spgemm.txt.
// With SPMM_HARNESS_TRACE=<dir> every array of the pipeline, the per-block formats included (which
// the reference computes and drops, :221 by-value parameters, :363), is written to
// <dir>/<name>/<array>.npy (NumPy format 1.0, little endian, C order).
// With SPMM_HARNESS_CACHE=1 the entries read from <name>.mtx are also written to <name>.mtx.bin (sizes, a
// fingerprint of the text, then the zero-based row and column ids), and a later run whose text has the same
// fingerprint (size, first and last 64 KiB) reads that file instead of the text: the reference reads the text
// with `fin >>` on every run (:47-104), which for a graph of 2.6e8 entries is minutes.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/spmm_hip.h"

namespace {

std::string strip_name(const std::string& s) {   // serial_newblock_clock.cpp:36-45
    std::string out;
    for (char ch : s) {
        if (ch == '.') break;
        out.push_back(ch);
    }
    return out;
}

// The bytes of a file, padded so that the library may read whole 16-byte words.
bool read_file(const std::string& path, std::vector<char>& buf, size_t& bytes) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 0) { std::fclose(f); return false; }
    bytes = (size_t)sz;
    buf.assign(bytes + 16, ' ');
    const bool ok = bytes == 0 || std::fread(buf.data(), 1, bytes, f) == bytes;
    std::fclose(f);
    return ok;
}

// One array as a .npy file (format 1.0: magic, version, header length, a dict padded to 64 bytes).
template <typename T>
bool write_npy(const std::string& path, const T* a, size_t n, const char* descr) {
    std::string h = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (" + std::to_string(n) + ",), }";
    while ((10 + h.size() + 1) % 64 != 0) h.push_back(' ');
    h.push_back('\n');
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    const unsigned char len[2] = {(unsigned char)(h.size() & 255), (unsigned char)(h.size() >> 8)};
    bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(len, 1, 2, f) == 2 && std::fwrite(h.data(), 1, h.size(), f) == h.size();
    ok = ok && (n == 0 || std::fwrite(a, sizeof(T), n, f) == n);
    return std::fclose(f) == 0 && ok;
}

// ---- binary side-cache of the entries of a .mtx file -----------------------------------------------------
struct CacheHeader {
    char magic[8];            // "SPMMBIN1"
    int32_t nrow, ncol;
    int64_t nnz;
    uint64_t text_bytes, text_fingerprint;
};

// FNV-1a-64 over the size of the file and its first and last 64 KiB: what is compared instead of re-reading the text
bool text_fingerprint(const std::string& path, uint64_t* bytes, uint64_t* fp) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    if (sz < 0) { std::fclose(f); return false; }
    const size_t piece = 64 * 1024;
    std::vector<unsigned char> buf(2 * piece);
    size_t got = 0;
    std::fseek(f, 0, SEEK_SET);
    got += std::fread(buf.data(), 1, std::min<size_t>(piece, (size_t)sz), f);
    if ((size_t)sz > piece) {
        const size_t tail = std::min<size_t>(piece, (size_t)sz - piece);
        std::fseek(f, (long)((size_t)sz - tail), SEEK_SET);
        got += std::fread(buf.data() + got, 1, tail, f);
    }
    std::fclose(f);
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 8; i++) h = (h ^ (((uint64_t)sz >> (8 * i)) & 255u)) * 1099511628211ull;
    for (size_t i = 0; i < got; i++) h = (h ^ buf[i]) * 1099511628211ull;
    *bytes = (uint64_t)sz;
    *fp = h;
    return true;
}

bool load_cache(const std::string& bin, uint64_t bytes, uint64_t fp, int* nrow, int* ncol, std::vector<int>& row, std::vector<int>& col) {
    FILE* f = std::fopen(bin.c_str(), "rb");
    if (!f) return false;
    CacheHeader h;
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "SPMMBIN1", 8) == 0 && h.text_bytes == bytes &&
              h.text_fingerprint == fp && h.nnz >= 0 && h.nnz < (1ll << 31) && h.nrow >= 0 && h.ncol >= 0;
    if (ok) {
        row.resize((size_t)h.nnz);
        col.resize((size_t)h.nnz);
        ok = (h.nnz == 0 || (std::fread(row.data(), 4, (size_t)h.nnz, f) == (size_t)h.nnz && std::fread(col.data(), 4, (size_t)h.nnz, f) == (size_t)h.nnz));
        *nrow = h.nrow;
        *ncol = h.ncol;
    }
    std::fclose(f);
    return ok;
}

void save_cache(const std::string& bin, uint64_t bytes, uint64_t fp, int nrow, int ncol, const std::vector<int>& row, const std::vector<int>& col) {
    const std::string tmp = bin + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;                           // a read-only input directory: run without the cache
    CacheHeader h;
    std::memcpy(h.magic, "SPMMBIN1", 8);
    h.nrow = nrow; h.ncol = ncol; h.nnz = (int64_t)row.size(); h.text_bytes = bytes; h.text_fingerprint = fp;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    ok = ok && (row.empty() || (std::fwrite(row.data(), 4, row.size(), f) == row.size() && std::fwrite(col.data(), 4, col.size(), f) == col.size()));
    ok = (std::fclose(f) == 0) && ok;
    if (ok) std::rename(tmp.c_str(), bin.c_str());
    else std::remove(tmp.c_str());
}

#define CK(call)                                                              \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != 0) {                                                       \
            std::cerr << #call << " failed: " << spmm_last_error() << "\n";   \
            return false;                                                     \
        }                                                                     \
    } while (0)

bool process(spmm_ctx* ctx, const std::string& mname, std::ofstream& outfile, std::ofstream* spg) {
    const std::string path = "mat/mtx/" + mname + "/" + mname + ".mtx";
    int nrow = 0, ncol = 0;
    std::vector<int> row, col;
    auto t0 = std::chrono::high_resolution_clock::now();
    const char* cache_env = std::getenv("SPMM_HARNESS_CACHE");
    const bool use_cache = cache_env && std::atoi(cache_env) != 0;
    uint64_t text_bytes = 0, text_fp = 0;
    bool from_cache = false;
    if (use_cache && text_fingerprint(path, &text_bytes, &text_fp))
        from_cache = load_cache(path + ".bin", text_bytes, text_fp, &nrow, &ncol, row, col);
    if (!from_cache) {
        // MatrixMarket text -> entries (serial_newblock_clock.cpp:47-104): header on the host,
        // the entries on the GPU
        std::vector<char> text;
        size_t bytes = 0, data_offset = 0;
        if (!read_file(path, text, bytes)) {
            std::cerr << "cannot read " << path << "\n";
            return false;
        }
        int64_t nz = 0;
        int per = 2;
        CK(spmm_mtx_header(text.data(), bytes, &nrow, &ncol, &nz, &data_offset, &per));
        row.resize((size_t)nz);
        col.resize((size_t)nz);
        CK(spmm_mtx_entries(ctx, text.data(), bytes, data_offset, per, nz, row.data(), col.data()));
        if (use_cache && text_bytes == bytes) save_cache(path + ".bin", text_bytes, text_fp, nrow, ncol, row, col);
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    const int nnz = (int)row.size();
    std::vector<int> indptr((size_t)nrow + 1), indices((size_t)nnz);
    std::vector<double> data((size_t)nnz);
    CK(spmm_csr_from_coo(ctx, nrow, ncol, nnz, row.data(), col.data(), indptr.data(), indices.data(), data.data()));
    const double ms_csr = spmm_last_kernel_ms(ctx);
    spmm_csr A{nrow, ncol, nnz, indptr.data(), indices.data(), data.data()};

    // caller-allocated outputs, like main()'s seq / bsize_list / seq_offset (:543-553), but sized
    // so that no block count overflows them
    const int cap = nrow + 1;
    std::vector<int> seq_input((size_t)nnz + 1), rseq((size_t)nrow + 1), bsize((size_t)cap + 1), seq_offset((size_t)cap + 1);
    spmm_trace tr;
    std::memset(&tr, 0, sizeof tr);
    tr.cap_blocks = cap;
    tr.seq_input = seq_input.data();
    tr.rseq = rseq.data();
    tr.bsize_list = bsize.data();
    tr.seq_offset = seq_offset.data();
    // the on-disk trace wants every array (sizes as documented for spmm_trace)
    const char* trace_dir = std::getenv("SPMM_HARNESS_TRACE");
    const size_t npan = (size_t)nrow / 8 + 3 * (size_t)cap + 8;
    std::vector<int> seq_bitmap, r_indptr, r_indices, panel_ptr, panel_list, spv8_list, out_seq_all, b_indptr, b_colidx_orig, b_local, keys;
    std::vector<double> r_data, b_data;
    if (trace_dir) {
        seq_bitmap.resize((size_t)nrow + 1); r_indptr.resize((size_t)nrow + 1); r_indices.resize((size_t)nnz + 1); r_data.resize((size_t)nnz + 1);
        panel_ptr.resize((size_t)cap + 1); panel_list.resize(npan); spv8_list.resize(npan); out_seq_all.resize((size_t)nrow + 1);
        b_indptr.resize((size_t)nrow + cap + 1); b_colidx_orig.resize((size_t)nnz + 1); b_local.resize((size_t)nnz + 1);
        b_data.resize((size_t)nnz + 1); keys.resize((size_t)nnz + 1);
        tr.seq_bitmap = seq_bitmap.data(); tr.r_indptr = r_indptr.data(); tr.r_indices = r_indices.data(); tr.r_data = r_data.data();
        tr.panel_ptr = panel_ptr.data(); tr.panel_list = panel_list.data(); tr.spv8_list = spv8_list.data();
        tr.out_seq_all = out_seq_all.data(); tr.b_indptr = b_indptr.data(); tr.b_colidx_orig = b_colidx_orig.data();
        tr.b_local = b_local.data(); tr.b_data = b_data.data(); tr.keys = keys.data();
    }
    auto t2 = std::chrono::high_resolution_clock::now();
    CK(spmm_trace_formats(ctx, &A, 2048, 512 * 1024 / 8, &tr));
    auto t3 = std::chrono::high_resolution_clock::now();
    const double ms_gpu = spmm_last_kernel_ms(ctx);
    const double ms_wall = std::chrono::duration<double, std::milli>(t3 - t2).count();
    const double ms_read = std::chrono::duration<double, std::milli>(t1 - t0).count();

    const char* tmode = std::getenv("SPMM_HARNESS_TIME");
    outfile << mname << " " << ((tmode && std::strcmp(tmode, "gpu") == 0) ? ms_gpu : ms_wall) << "ms" << std::endl;
    std::cout << "----name:" << mname << "----" << std::endl;

    if (trace_dir) {
        const std::string d = std::string(trace_dir) + "/" + mname;
        std::error_code ec;
        std::filesystem::create_directories(d, ec);
        const size_t nb = (size_t)tr.bnum, nk = (size_t)tr.n_keys, z = (size_t)nnz, nr = (size_t)nrow;
        bool ok = !ec;
        ok = ok && write_npy(d + "/seq_bitmap.npy", seq_bitmap.data(), nr, "<i4") && write_npy(d + "/r_indptr.npy", r_indptr.data(), nr + 1, "<i4");
        ok = ok && write_npy(d + "/r_indices.npy", r_indices.data(), z, "<i4") && write_npy(d + "/r_data.npy", r_data.data(), z, "<f8");
        ok = ok && write_npy(d + "/bsize_list.npy", bsize.data(), nb + 1, "<i4") && write_npy(d + "/seq_offset.npy", seq_offset.data(), nb + 1, "<i4");
        ok = ok && write_npy(d + "/panel_ptr.npy", panel_ptr.data(), nb + 1, "<i4");
        ok = ok && write_npy(d + "/panel_list.npy", panel_list.data(), (size_t)tr.n_panel_entries, "<i4");
        ok = ok && write_npy(d + "/spv8_list.npy", spv8_list.data(), (size_t)tr.n_spv8, "<i4");
        ok = ok && write_npy(d + "/out_seq_all.npy", out_seq_all.data(), nr, "<i4") && write_npy(d + "/b_indptr.npy", b_indptr.data(), nr + nb, "<i4");
        ok = ok && write_npy(d + "/b_colidx_orig.npy", b_colidx_orig.data(), z, "<i4") && write_npy(d + "/b_local.npy", b_local.data(), z, "<i4");
        ok = ok && write_npy(d + "/b_data.npy", b_data.data(), z, "<f8") && write_npy(d + "/keys.npy", keys.data(), nk, "<i4");
        ok = ok && write_npy(d + "/seq_input.npy", seq_input.data(), nk, "<i4") && write_npy(d + "/rseq.npy", rseq.data(), nr, "<i4");
        if (!ok) {
            std::cerr << "cannot write the trace under " << d << "\n";
            return false;
        }
    }

    if (spg) {
        std::vector<int64_t> c_ip((size_t)nrow + 1);
        int64_t nnz_c = 0, products = 0;
        CK(spmm_spgemm_symbolic(ctx, &A, &A, c_ip.data(), &nnz_c, &products));
        const double ms_sym = spmm_last_kernel_ms(ctx);
        double ms_num = 0.0, sum = 0.0;
        uint64_t hash = 1469598103934665603ull;
        const int64_t chunk = 64ll << 20;   // entries of C per numeric call
        std::vector<int> ci;
        std::vector<double> cd;
        for (int r0 = 0; r0 < nrow;) {
            int r1 = r0 + 1;
            while (r1 < nrow && c_ip[(size_t)r1 + 1] - c_ip[(size_t)r0] <= chunk) r1++;
            const int64_t nz = c_ip[(size_t)r1] - c_ip[(size_t)r0];
            ci.resize((size_t)nz + 1);
            cd.resize((size_t)nz + 1);
            CK(spmm_spgemm_numeric(ctx, &A, &A, c_ip.data(), r0, r1, ci.data(), cd.data()));
            ms_num += spmm_last_kernel_ms(ctx);
            for (int64_t t = 0; t < nz; t++) {
                hash = (hash ^ (uint32_t)ci[(size_t)t]) * 1099511628211ull;
                sum += cd[(size_t)t];
            }
            r0 = r1;
        }
        char line[1024];
        std::snprintf(line, sizeof line,
                      "{\"name\": \"%s\", \"rows\": %d, \"nnz\": %d, \"entries_from\": \"%s\", \"read_ms\": %.3f, \"csr_build_gpu_ms\": %.3f, "
                      "\"preprocess_gpu_ms\": %.3f, \"preprocess_wall_ms\": %.3f, \"blocks\": %d, \"products\": %lld, "
                      "\"nnz_c\": %lld, \"symbolic_gpu_ms\": %.3f, \"numeric_gpu_ms\": %.3f, \"gflops_numeric\": %.2f, "
                      "\"c_indices_fnv1a64\": \"%016llx\", \"c_value_sum\": %.17g}",
                      mname.c_str(), nrow, nnz, from_cache ? "cache" : "text", ms_read, ms_csr, ms_gpu, ms_wall, tr.bnum, (long long)products,
                      (long long)nnz_c, ms_sym, ms_num, ms_num > 0 ? 2.0 * products / (ms_num * 1e-3) / 1e9 : 0.0,
                      (unsigned long long)hash, sum);
        *spg << line << std::endl;
    }
    return true;
}

}  // namespace

int main() {
    std::vector<std::string> mlist;
    {
        std::ifstream file("matrix.txt");
        std::string s;
        while (std::getline(file, s)) mlist.push_back(s);
    }
    std::ofstream outfile("result.txt");
    if (!outfile) {
        std::cout << "result.txt not exist";
        return 1;
    }
    spmm_ctx* ctx = nullptr;
    int dev = 0;
    if (const char* e = std::getenv("SPMM_DEVICE")) dev = std::atoi(e);
    if (spmm_ctx_create(dev, &ctx) != 0) {
        std::cerr << "no usable MI355X: " << spmm_last_error() << "\n";
        return 2;
    }
    std::ofstream spg_file;
    std::ofstream* spg = nullptr;
    if (const char* e = std::getenv("SPMM_HARNESS_SPGEMM")) {
        if (std::atoi(e) != 0) {
            spg_file.open("result_spgemm.txt");
            spg = &spg_file;
        }
    }
    int rc = 0;
    for (const std::string& line : mlist) {
        const std::string mname = strip_name(line);
        if (mname.empty()) continue;
        if (!process(ctx, mname, outfile, spg)) rc = 3;
    }
    spmm_ctx_destroy(ctx);
    return rc;
}
