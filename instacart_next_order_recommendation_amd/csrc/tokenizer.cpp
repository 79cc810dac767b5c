// tokenizer.cpp — native, batched BERT WordPiece tokenisation (host side of icrec_encode's input).
//
// The reference tokenises inside SentenceTransformer.encode (serve_recommendations.py:213,:246)
// through transformers' BertTokenizer -> the Rust `tokenizers` crate (uv.lock:3841).  This is a C++
// restatement of that pipeline so a serving process can tokenise request batches on worker threads
// without the GIL:
//   special-token extraction ([PAD] [UNK] [CLS] [SEP] [MASK] matched verbatim in the raw text)
//   BertNormalizer   clean_text (drop NUL / U+FFFD / control, whitespace -> ' '), spaces around CJK
//                    ideographs, strip accents (NFD, drop Mn) when lower-casing, lower-case
//   BertPreTokenizer split on whitespace, isolate punctuation (ASCII punctuation + Unicode P*)
//   WordPiece        greedy longest-match-first, "##" continuation, > 100 chars or no match -> [UNK]
//   template         [CLS] tokens [SEP], truncated to max_len (tokens cut on the right)
// Unicode data: unicode_tables.h (generated from Python's unicodedata).  Known difference from the Rust
// crate: unassigned code points are kept (-> [UNK]) instead of being dropped as "other".
#include <sched.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/icrec.h"
#include "unicode_tables.h"

namespace icrec {
void set_error(const char* fmt, ...);
}
using icrec::set_error;

namespace {

typedef std::u32string U32;

bool in_ranges(const CpRange* r, int n, uint32_t cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (cp < r[mid].lo) hi = mid - 1;
        else if (cp > r[mid].hi) lo = mid + 1;
        else return true;
    }
    return false;
}
const CpMap* find_map(const CpMap* m, int n, uint32_t cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (cp < m[mid].cp) hi = mid - 1;
        else if (cp > m[mid].cp) lo = mid + 1;
        else return &m[mid];
    }
    return nullptr;
}

inline bool is_ws(uint32_t c) { return c == 0x2028 || c == 0x2029 || in_ranges(kWhitespace, kWhitespace_n, c); }
inline bool is_ctrl(uint32_t c) { return in_ranges(kControl, kControl_n, c); }
inline bool is_punct(uint32_t c) { return in_ranges(kPunct, kPunct_n, c); }
inline bool is_mn(uint32_t c) { return in_ranges(kMn, kMn_n, c); }
inline bool is_cjk(uint32_t c) {
    return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF) || (c >= 0x20000 && c <= 0x2A6DF) ||
           (c >= 0x2A700 && c <= 0x2B73F) || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B820 && c <= 0x2CEAF) ||
           (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}

void decode_utf8(const char* s, size_t n, U32& out) {
    out.clear();
    size_t i = 0;
    while (i < n) {
        unsigned char c = (unsigned char)s[i];
        uint32_t cp;
        int len;
        if (c < 0x80) { cp = c; len = 1; }
        else if ((c >> 5) == 6) { cp = c & 0x1F; len = 2; }
        else if ((c >> 4) == 14) { cp = c & 0x0F; len = 3; }
        else if ((c >> 3) == 30) { cp = c & 0x07; len = 4; }
        else { out.push_back(0xFFFD); ++i; continue; }
        if (i + len > n) { out.push_back(0xFFFD); break; }
        bool ok = true;
        for (int j = 1; j < len; ++j) {
            unsigned char d = (unsigned char)s[i + j];
            if ((d >> 6) != 2) { ok = false; break; }
            cp = (cp << 6) | (d & 0x3F);
        }
        if (!ok) { out.push_back(0xFFFD); ++i; continue; }
        out.push_back(cp);
        i += len;
    }
}
void append_utf8(std::string& s, uint32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
        s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

// canonical decomposition with combining marks (Mn) dropped
void nfd_strip(uint32_t c, U32& out) {
    if (c >= 0xAC00 && c <= 0xD7A3) {  // Hangul syllable -> L V (T)
        uint32_t s = c - 0xAC00;
        out.push_back(0x1100 + s / 588);
        out.push_back(0x1161 + (s % 588) / 28);
        if (s % 28) out.push_back(0x11A7 + s % 28);
        return;
    }
    if (const CpMap* m = find_map(kNfd, kNfd_n, c)) {
        for (uint32_t i = 0; i < m->len; ++i)
            if (!is_mn(kNfd_pool[m->off + i])) out.push_back(kNfd_pool[m->off + i]);
    } else if (!is_mn(c)) {
        out.push_back(c);
    }
}
void lower(uint32_t c, U32& out) {
    if (c < 0x80) { out.push_back((c >= 'A' && c <= 'Z') ? c + 32 : c); return; }
    if (const CpMap* m = find_map(kLower, kLower_n, c))
        for (uint32_t i = 0; i < m->len; ++i) out.push_back(kLower_pool[m->off + i]);
    else out.push_back(c);
}

}  // namespace

struct icrec_tokenizer {
    std::unordered_map<std::string, int32_t> vocab;
    bool do_lower = true;
    int max_len = 256;
    int32_t unk = 100, cls = 101, sep = 102;
    std::vector<std::pair<std::string, int32_t>> specials;

    void wordpiece(const U32& w, std::vector<int32_t>& out, std::string& buf) const {
        if (w.size() > 100) { out.push_back(unk); return; }
        const size_t first = out.size();
        size_t start = 0;
        while (start < w.size()) {
            size_t end = w.size();
            int32_t id = -1;
            while (start < end) {
                buf.clear();
                if (start > 0) buf = "##";
                for (size_t i = start; i < end; ++i) append_utf8(buf, w[i]);
                auto it = vocab.find(buf);
                if (it != vocab.end()) { id = it->second; break; }
                --end;
            }
            if (id < 0) { out.resize(first); out.push_back(unk); return; }
            out.push_back(id);
            start = end;
        }
    }

    // normalise + pre-tokenise + wordpiece one stretch of ordinary text
    void segment(const char* s, size_t n, std::vector<int32_t>& out) const {
        U32 raw, norm, word, tmp;
        std::string buf;
        decode_utf8(s, n, raw);
        norm.reserve(raw.size() + 8);
        for (uint32_t c : raw) {
            if (c == 0 || c == 0xFFFD || is_ctrl(c)) continue;
            if (is_ws(c)) { norm.push_back(' '); continue; }
            if (is_cjk(c)) { norm.push_back(' '); norm.push_back(c); norm.push_back(' '); continue; }
            if (do_lower) {
                tmp.clear();
                nfd_strip(c, tmp);
                for (uint32_t d : tmp) lower(d, norm);
            } else {
                norm.push_back(c);
            }
        }
        auto flush = [&]() { if (!word.empty()) { wordpiece(word, out, buf); word.clear(); } };
        for (uint32_t c : norm) {
            if (c == ' ' || is_ws(c)) { flush(); }
            else if (is_punct(c)) { flush(); word.push_back(c); flush(); }
            else word.push_back(c);
        }
        flush();
    }

    void encode(const char* text, std::vector<int32_t>& out) const {
        std::vector<int32_t> body;
        const size_t n = strlen(text);
        size_t pos = 0, seg = 0;
        while (pos < n) {  // special tokens are matched verbatim in the raw text, never split
            bool hit = false;
            if (text[pos] == '[') {
                for (auto& sp : specials)
                    if (n - pos >= sp.first.size() && memcmp(text + pos, sp.first.data(), sp.first.size()) == 0) {
                        if (pos > seg) segment(text + seg, pos - seg, body);
                        body.push_back(sp.second);
                        pos += sp.first.size();
                        seg = pos;
                        hit = true;
                        break;
                    }
            }
            if (!hit) ++pos;
        }
        if (n > seg) segment(text + seg, n - seg, body);
        const size_t keep = std::min(body.size(), (size_t)std::max(0, max_len - 2));
        out.push_back(cls);
        out.insert(out.end(), body.begin(), body.begin() + keep);
        out.push_back(sep);
    }
};

extern "C" {

int icrec_tokenizer_create(const char* vocab_path, int do_lower_case, int max_len, icrec_tokenizer** out) {
    if (!vocab_path || !out) { set_error("icrec_tokenizer_create: NULL argument"); return ICREC_EINVAL; }
    if (max_len < 2) { set_error("icrec_tokenizer_create: max_len must be >= 2"); return ICREC_EINVAL; }
    std::ifstream f(vocab_path);
    if (!f) { set_error("icrec_tokenizer_create: cannot open %s", vocab_path); return ICREC_EINVAL; }
    icrec_tokenizer* t = new icrec_tokenizer();
    t->do_lower = do_lower_case != 0;
    t->max_len = max_len;
    std::string line;
    int32_t id = 0;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        t->vocab.emplace(line, id++);  // first occurrence wins, ids are line numbers
    }
    auto need = [&](const char* tok, int32_t& dst) {
        auto it = t->vocab.find(tok);
        if (it == t->vocab.end()) return false;
        dst = it->second;
        return true;
    };
    if (!need("[UNK]", t->unk) || !need("[CLS]", t->cls) || !need("[SEP]", t->sep)) {
        delete t;
        set_error("icrec_tokenizer_create: %s lacks [UNK]/[CLS]/[SEP]", vocab_path);
        return ICREC_EINVAL;
    }
    for (const char* sp : {"[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"}) {
        auto it = t->vocab.find(sp);
        if (it != t->vocab.end()) t->specials.emplace_back(sp, it->second);
    }
    *out = t;
    return ICREC_OK;
}

// CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a 16-CPU container on a
// 256-CPU host reports 256 hardware threads; spawning that many only adds scheduling overhead).
static int usable_cpus() {
    static int cached = 0;
    if (cached > 0) return cached;
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : 1 << 30, CPU_COUNT(&set));
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string quota;
    long long period = 0;
    if (f && (f >> quota >> period) && quota != "max" && period > 0) {
        const long long q = atoll(quota.c_str());
        if (q > 0) n = std::min(n, (int)((q + period - 1) / period));
    }
    cached = std::max(1, n);
    return cached;
}

int icrec_tokenizer_destroy(icrec_tokenizer* t) { delete t; return ICREC_OK; }

int32_t icrec_tokenizer_vocab_size(const icrec_tokenizer* t) { return t ? (int32_t)t->vocab.size() : 0; }

int icrec_tokenize(const icrec_tokenizer* t, const char* const* texts, int32_t n, int32_t* out_ids, int64_t cap,
                   int32_t* out_cu, int32_t n_threads) {
    if (!t || !texts || !out_cu || n < 0 || (cap > 0 && !out_ids)) { set_error("icrec_tokenize: bad argument"); return ICREC_EINVAL; }
    int nt = n_threads > 0 ? n_threads : usable_cpus();
    nt = std::max(1, std::min(nt, std::max(1, n / 64)));  // below ~64 texts per thread the spawn costs more
    std::vector<std::vector<int32_t>> ids(nt), lens(nt);
    auto work = [&](int w) {
        const int i0 = (int)((int64_t)n * w / nt), i1 = (int)((int64_t)n * (w + 1) / nt);
        for (int i = i0; i < i1; ++i) {
            const size_t before = ids[w].size();
            t->encode(texts[i] ? texts[i] : "", ids[w]);
            lens[w].push_back((int32_t)(ids[w].size() - before));
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int w = 0; w < nt; ++w) th.emplace_back(work, w);
        for (auto& x : th) x.join();
    }
    int64_t total = 0;
    int32_t q = 0;
    out_cu[0] = 0;
    for (int w = 0; w < nt; ++w)
        for (int32_t l : lens[w]) { total += l; out_cu[++q] = (int32_t)total; }
    if (total > cap) {
        set_error("icrec_tokenize: need room for %lld ids, got %lld", (long long)total, (long long)cap);
        return ICREC_ENOMEM;  // out_cu is valid: the caller can size out_ids from out_cu[n] and retry
    }
    int64_t o = 0;
    for (int w = 0; w < nt; ++w) {
        if (!ids[w].empty()) memcpy(out_ids + o, ids[w].data(), ids[w].size() * sizeof(int32_t));
        o += (int64_t)ids[w].size();
    }
    return ICREC_OK;
}

}  // extern "C"
