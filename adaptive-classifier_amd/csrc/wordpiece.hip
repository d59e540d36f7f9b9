// On-device BERT WordPiece tokenisation: the `self.tokenizer(texts, max_length, truncation=True,
// padding=True)` call of /root/reference/src/adaptive_classifier/classifier.py:1259-1265 for BERT-family vocabularies
// (transformers BertTokenizer = tokenizers BertNormalizer + BertPreTokenizer + WordPiece), ASCII texts.
//
//   normaliser      drop NUL / control bytes (0x00-0x08, 0x0b, 0x0c, 0x0e-0x1f, 0x7f), map \t \n \r to ' ', lower-case
//                   A-Z when the vocabulary is uncased (BertNormalizer clean_text / lowercase; accent stripping and CJK
//                   spacing cannot trigger on ASCII)
//   pre-tokeniser   split on spaces; every ASCII punctuation byte (33-47, 58-64, 91-96, 123-126) is its own word
//   WordPiece       greedy longest-match-first per word, continuation pieces looked up with the "##" prefix, words of
//                   more than 100 characters or with an unmatched remainder -> [UNK]
//   post-processor  [CLS] pieces[: max_length - 2] [SEP], padded with id 0; attention mask
// Texts with non-ASCII bytes or literal special-token strings are tokenised by the host tokenizer (the Python wrapper
// routes them); everything else never leaves the GPU: the ids feed ac_bert_encode_cls directly.
//
// One wave per text (texts of up to 4096 bytes; their cleaned bytes live in LDS).  Step 1 compacts the cleaned bytes
// (ballot + prefix popcount); step 2 walks them in 64-byte
// chunks: lanes standing on a word start count their word's pieces, a wave prefix sum gives the output slots, and
// the lanes match again to write the ids (matching twice is cheaper than staging variable-length piece lists).
// Vocabulary: open-addressing hash table (FNV-1a 64 over the piece bytes, "##" folded into the initial state for
// continuation pieces) with byte-exact verification against the vocabulary blob.
#include "common.h"

namespace {

constexpr uint64_t kFnvOffset = 0xcbf29ce484222325ull, kFnvPrime = 0x100000001b3ull;
constexpr int kMaxWordChars = 100;          // tokenizers WordPiece max_input_chars_per_word

struct WpTable {
    const uint64_t* keys;      // [slots] full hash, 0 = empty
    const uint32_t* offs;      // [slots] offset of the piece's bytes in blob (without the "##" prefix)
    const int32_t* ids;        // [slots] token id; bit 30 set = continuation piece ("##...")
    const uint16_t* lens;      // [slots] byte length (without prefix)
    const uint8_t* blob;
    uint32_t mask;             // slots - 1
    int max_piece;             // longest piece in the vocabulary (bytes, without prefix)
    int unk_id, cls_id, sep_id, pad_id;
    int lower;
};

__host__ __device__ __forceinline__ uint64_t fnv_step(uint64_t h, uint8_t c) { return (h ^ c) * kFnvPrime; }

// id of the piece word[s, e) (continuation iff cont), or -1
__device__ __forceinline__ int wp_lookup(const WpTable& t, uint64_t h, const uint8_t* p, int len, bool cont) {
    if (h == 0) h = 1;
    uint32_t slot = (uint32_t)(h ^ (h >> 32)) & t.mask;
    for (;;) {
        const uint64_t k = t.keys[slot];
        if (k == 0) return -1;
        if (k == h && t.lens[slot] == len) {
            const int32_t v = t.ids[slot];
            if (((v >> 30) & 1) == (cont ? 1 : 0)) {
                const uint8_t* b = t.blob + t.offs[slot];
                bool eq = true;
                for (int i = 0; i < len; ++i) eq &= b[i] == p[i];
                if (eq) return v & 0x3fffffff;
            }
        }
        slot = (slot + 1) & t.mask;
    }
}

// Greedy longest-match-first over one word.  out == nullptr: returns the number of pieces (1 for [UNK]).  Otherwise
// `expect` is that count and up to `room` ids are written (a word whose count is 1 is [UNK] unless its first match
// covers it whole; a count > 1 means every piece matches, so ids can be written as they are found).
__device__ int wp_word(const WpTable& t, const uint8_t* w, int len, int64_t* out, int room, int expect) {
    if (len > kMaxWordChars) { if (out && room > 0) out[0] = t.unk_id; return 1; }
    int n = 0, s = 0;
    while (s < len) {
        // forward scan from s: remember the longest end with a vocabulary hit
        uint64_t h = kFnvOffset;
        if (s > 0) { h = fnv_step(h, '#'); h = fnv_step(h, '#'); }
        int best_e = -1, best_id = -1;
        const int emax = len - s < t.max_piece ? len : s + t.max_piece;
        for (int e = s; e < emax; ++e) {
            h = fnv_step(h, w[e]);
            const int id = wp_lookup(t, h, w + s, e + 1 - s, s > 0);
            if (id >= 0) { best_e = e + 1; best_id = id; }
        }
        if (best_e < 0) { if (out && room > 0) out[0] = t.unk_id; return 1; }      // whole word -> [UNK]
        if (out && expect == 1) {                                                      // single slot: the word or [UNK]
            if (room > 0) out[0] = best_e == len ? best_id : t.unk_id;
            return 1;
        }
        if (out && n < room) out[n] = best_id;
        ++n;
        s = best_e;
    }
    return n;
}

__device__ __forceinline__ int cls_of(uint8_t c) {        // cleaned byte: 1 space, 2 punctuation, 3 word character
    if (c == ' ') return 1;
    if ((c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126)) return 2;
    return 3;
}

constexpr int kWpMaxBytes = 4096;           // cleaned bytes of one text kept in LDS (longer texts: host tokenizer)

__global__ __launch_bounds__(256) void wordpiece_kernel(const uint8_t* __restrict__ text, const int32_t* __restrict__ offs, int b,
                                                        WpTable t, int max_len, int64_t* __restrict__ ids,
                                                        int64_t* __restrict__ mask, int32_t* __restrict__ lens) {
    __shared__ uint8_t clean[4][kWpMaxBytes];
    const int lane = threadIdx.x & 63;
    const int tx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tx >= b) return;
    const int o0 = offs[tx];
    int o1 = offs[tx + 1];
    if (o1 - o0 > kWpMaxBytes) o1 = o0 + kWpMaxBytes;               // (the wrapper never sends longer texts)
    uint8_t* c = clean[threadIdx.x >> 6];
    // ---- 1. clean + lower-case + compact ----
    int n = 0;
    for (int base = o0; base < o1; base += 64) {
        const int p = base + lane;
        uint8_t ch = p < o1 ? text[p] : 0;
        bool keep = p < o1;
        if (ch == '\t' || ch == '\n' || ch == '\r') ch = ' ';
        else if (ch < 0x20 || ch == 0x7f) keep = false;
        if (t.lower && ch >= 'A' && ch <= 'Z') ch += 32;
        const uint64_t m = __ballot(keep);
        if (keep) c[n + __popcll(m & ((1ull << lane) - 1))] = ch;
        n += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 2. words -> pieces ----
    int64_t* row = ids + (size_t)tx * max_len;
    const int room_total = max_len - 2;
    int total = 0;
    for (int base = 0; base < n && total < room_total; base += 64) {
        const int p = base + lane;
        int wlen = 0;
        if (p < n) {
            const int k = cls_of(c[p]);
            const bool start = k == 2 || (k == 3 && (p == 0 || cls_of(c[p - 1]) != 3));
            if (start) {
                wlen = 1;
                if (k == 3) while (p + wlen < n && cls_of(c[p + wlen]) == 3 && wlen <= kMaxWordChars) ++wlen;
            }
        }
        int cnt = wlen ? wp_word(t, c + p, wlen, nullptr, 0, 0) : 0;
        // exclusive wave prefix sum of cnt
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const int slot = total + incl - cnt;
        if (wlen && slot < room_total) wp_word(t, c + p, wlen, row + 1 + slot, room_total - slot, cnt);
        total += __shfl(incl, 63);
    }
    if (total > room_total) total = room_total;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 3. specials, padding, mask ----
    const int L = total + 2;
    if (lane == 0) { row[0] = t.cls_id; row[L - 1] = t.sep_id; lens[tx] = L; }
    int64_t* mrow = mask + (size_t)tx * max_len;
    for (int i = lane; i < max_len; i += 64) {
        if (i >= L) row[i] = t.pad_id;
        mrow[i] = i < L ? 1 : 0;
    }
}

}  // namespace

extern "C" int ac_wordpiece_encode(const uint8_t* d_text, const int32_t* d_offsets, int b, const ac_wordpiece_vocab* v,
                                   int max_length, int64_t* d_ids, int64_t* d_mask, int32_t* d_lens, ac_stream_t stream) {
    AC_REQUIRE(v && d_offsets && d_ids && d_mask && d_lens && b >= 0 && max_length >= 2, AC_EINVAL, "wordpiece: bad arguments");
    AC_REQUIRE(v->slots >= 2 && (v->slots & (v->slots - 1)) == 0 && v->keys && v->offs && v->ids && v->lens && v->blob, AC_EINVAL,
               "wordpiece: vocabulary table must have a power-of-two slot count and all arrays");
    if (b == 0) return AC_OK;
    AC_REQUIRE(d_text, AC_EINVAL, "wordpiece: text is NULL");
    WpTable t;
    t.keys = v->keys; t.offs = v->offs; t.ids = v->ids; t.lens = v->lens; t.blob = v->blob; t.mask = (uint32_t)(v->slots - 1);
    t.max_piece = v->max_piece_bytes; t.unk_id = v->unk_id; t.cls_id = v->cls_id; t.sep_id = v->sep_id; t.pad_id = v->pad_id;
    t.lower = v->lower_case;
    hipLaunchKernelGGL(wordpiece_kernel, dim3((b + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_text, d_offsets, b, t, max_length,
                       d_ids, d_mask, d_lens);
    AC_LAUNCH_CHECK();
    return AC_OK;
}

// host helper shared with the Python wrapper's table builder (and tests): the hash a piece is stored under
extern "C" uint64_t ac_wordpiece_hash(const uint8_t* bytes, int len, int continuation) {
    uint64_t h = kFnvOffset;
    if (continuation) { h = fnv_step(h, '#'); h = fnv_step(h, '#'); }
    for (int i = 0; i < len; ++i) h = fnv_step(h, bytes[i]);
    return h == 0 ? 1 : h;
}
