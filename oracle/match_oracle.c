/*
 * CPU ORACLE (test infrastructure, NOT product code): integer hot loops of
 * linemodLevelup::Detector::match, restated in plain C with the same SSE2/SSSE3 operations the
 * reference uses.  Citations "LL.cpp:N" are to /root/reference/linemodLevelup/linemodLevelup.cpp.
 *
 * Pinned to the reference's own lines: tests/test_ref_pin.py compares every function here with oracle/_ref (the
 * cited LL.cpp ranges compiled unmodified against a cv::Mat buffer shim) on all reference fixtures.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Built by oracle/Makefile (or linemod_oracle.build_c) with the reference's flags: -O3 -Wall, no
 * -march (linemodLevelup/CMakeLists.txt:8).
 *
 * Memory layout of one (pyramid level, modality) block of linear memories:
 *     u8 LM[8 labels][T*T phases][(W/T)*(H/T)]  followed by a zero tail (see lm_tail_pad()).
 * The reference allocates one cv::Mat per label (LL.cpp:1223); features sitting at x==width or
 * y==height read past their phase row into the next one which this layout reproduces.
 */
#include <emmintrin.h>
#include <tmmintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t x, y; float sim; int32_t cls, tid; } mo_match;

/* ---- SIMILARITY_LUT (LL.cpp:1121): closed form, 8 orientations x (16 low-nibble + 16 high-nibble) */
static void make_lut(uint8_t lut[256])
{
    for (int ori = 0; ori < 8; ++ori)
        for (int half = 0; half < 2; ++half)
            for (int v = 0; v < 16; ++v) {
                int bits = half ? (v << 4) : v;
                int r = 0;
                if (bits & (1 << ori)) r = 4;
                else if ((bits & (1 << ((ori + 1) & 7))) || (bits & (1 << ((ori + 7) & 7)))) r = 1;
                lut[ori * 32 + half * 16 + v] = (uint8_t)r;
            }
}

void mo_similarity_lut(uint8_t *out256) { make_lut(out256); }

/* ---- spread (LL.cpp:1094-1109) with orUnaligned8u (LL.cpp:1026-1083) ---------------------- */
static void or_rows(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int width, int height)
{
    for (int r = 0; r < height; ++r) {
        int c = 0;
        for (; c < width - 15; c += 16) {
            __m128i v = _mm_loadu_si128((const __m128i *)(src + c));
            __m128i d = _mm_loadu_si128((const __m128i *)(dst + c));
            _mm_storeu_si128((__m128i *)(dst + c), _mm_or_si128(d, v));
        }
        for (; c < width; ++c) dst[c] |= src[c];
        src += src_stride;
        dst += dst_stride;
    }
}

static void spread(const uint8_t *src, uint8_t *dst, int W, int H, int T)
{
    memset(dst, 0, (size_t)W * H);
    for (int r = 0; r < T; ++r)
        for (int c = 0; c < T; ++c)
            or_rows(src + (size_t)r * W + c, W, dst, W, W - c, H - r);
}

/* ---- computeResponseMaps (LL.cpp:1134-1203): pshufb on both nibbles, max --------------------- */
__attribute__((target("ssse3")))
static void response_maps(const uint8_t *spr, uint8_t *maps /* [8][W*H] */, size_t n)
{
    uint8_t lut[256] __attribute__((aligned(16)));
    make_lut(lut);
    const __m128i mask = _mm_set1_epi8(15);
    for (int ori = 0; ori < 8; ++ori) {
        __m128i lo_t = _mm_load_si128((const __m128i *)(lut + 32 * ori));
        __m128i hi_t = _mm_load_si128((const __m128i *)(lut + 32 * ori + 16));
        uint8_t *m = maps + (size_t)ori * n;
        size_t i = 0;
        for (; i + 16 <= n; i += 16) {
            __m128i v = _mm_loadu_si128((const __m128i *)(spr + i));
            __m128i lo = _mm_and_si128(v, mask);
            __m128i hi = _mm_and_si128(_mm_srli_epi16(v, 4), mask);
            __m128i r = _mm_max_epu8(_mm_shuffle_epi8(lo_t, lo), _mm_shuffle_epi8(hi_t, hi));
            _mm_storeu_si128((__m128i *)(m + i), r);
        }
        for (; i < n; ++i) {
            uint8_t a = lut[32 * ori + (spr[i] & 15)], b = lut[32 * ori + 16 + (spr[i] >> 4)];
            m[i] = a > b ? a : b;
        }
    }
}

/* ---- linearize (LL.cpp:1215-1243) --------------------------------------------------------------- */
static void linearize(const uint8_t *resp, uint8_t *lin, int W, int H, int T)
{
    uint8_t *mem = lin;
    for (int rs = 0; rs < T; ++rs)
        for (int cs = 0; cs < T; ++cs)
            for (int r = rs; r < H; r += T) {
                const uint8_t *row = resp + (size_t)r * W;
                for (int c = cs; c < W; c += T) *mem++ = row[c];
            }
}

/* quantized (one-hot u8, W x H) -> LM[8][T*T][Wd*Hd] (+ caller-zeroed tail) */
int mo_build_linear_memories(const uint8_t *quantized, int W, int H, int T, uint8_t *out)
{
    size_t n = (size_t)W * H;
    if (n % 16 || W % T || H % T) return -1;      /* CV_Assert LL.cpp:1136, 1217-1218 */
    uint8_t *spr = (uint8_t *)malloc(n);
    uint8_t *maps = (uint8_t *)malloc(8 * n);
    if (!spr || !maps) { free(spr); free(maps); return -2; }
    spread(quantized, spr, W, H, T);
    response_maps(spr, maps, n);
    for (int l = 0; l < 8; ++l) linearize(maps + (size_t)l * n, out + (size_t)l * n, W, H, T);
    free(spr);
    free(maps);
    return 0;
}

void mo_spread(const uint8_t *q, uint8_t *dst, int W, int H, int T) { spread(q, dst, W, H, T); }

/* ---- accessLinearMemory (LL.cpp:1248-1271) ----------------------------------------------------- */
static inline const uint8_t *access_lm(const uint8_t *lm, int fx, int fy, int label, int T, int Wd, int Hd)
{
    size_t lm_len = (size_t)Wd * Hd;
    int grid = (fy % T) * T + (fx % T);
    return lm + ((size_t)label * T * T + grid) * lm_len + (size_t)(fy / T) * Wd + fx / T;
}

/* ---- similarity (LL.cpp:1284-1354): dst16[j] += lm[j], j < template_positions ------------------- */
static void similarity16(const uint8_t *lm, const int32_t *feat, int nfeat, int tw, int th,
                         uint16_t *dst, int W, int H, int T)
{
    int Wd = W / T, Hd = H / T;
    int wf = (tw - 1) / T + 1, hf = (th - 1) / T + 1;
    int span_x = Wd - wf, span_y = Hd - hf;
    int tp = span_y * Wd + span_x + 1;                 /* LL.cpp:1309 */
    memset(dst, 0, sizeof(uint16_t) * (size_t)Wd * Hd);
    const __m128i zero = _mm_setzero_si128();
    for (int i = 0; i < nfeat; ++i) {
        int fx = feat[3 * i], fy = feat[3 * i + 1], lab = feat[3 * i + 2];
        if (fx < 0 || fx >= W || fy < 0 || fy >= H) continue;   /* LL.cpp:1330 */
        const uint8_t *p = access_lm(lm, fx, fy, lab, T, Wd, Hd);
        int j = 0;
        for (; j < tp - 7; j += 8) {
            __m128i r = _mm_loadl_epi64((const __m128i *)(p + j));
            __m128i d = _mm_loadu_si128((const __m128i *)(dst + j));
            _mm_storeu_si128((__m128i *)(dst + j), _mm_add_epi16(d, _mm_unpacklo_epi8(r, zero)));
        }
        for (; j < tp; ++j) dst[j] = (uint16_t)(dst[j] + p[j]);
    }
}

/* ---- similarity_64 (LL.cpp:1450-1534): 8-bit accumulation when the first modality has < 64 ------ */
static void similarity8(const uint8_t *lm, const int32_t *feat, int nfeat, int tw, int th,
                        uint8_t *dst, int W, int H, int T)
{
    int Wd = W / T, Hd = H / T;
    int wf = (tw - 1) / T + 1, hf = (th - 1) / T + 1;
    int tp = (Hd - hf) * Wd + (Wd - wf) + 1;
    memset(dst, 0, (size_t)Wd * Hd);
    for (int i = 0; i < nfeat; ++i) {
        int fx = feat[3 * i], fy = feat[3 * i + 1], lab = feat[3 * i + 2];
        if (fx < 0 || fx >= W || fy < 0 || fy >= H) continue;
        const uint8_t *p = access_lm(lm, fx, fy, lab, T, Wd, Hd);
        int j = 0;
        for (; j < tp - 15; j += 16) {
            __m128i r = _mm_loadu_si128((const __m128i *)(p + j));
            __m128i d = _mm_loadu_si128((const __m128i *)(dst + j));
            _mm_storeu_si128((__m128i *)(dst + j), _mm_add_epi8(d, r));
        }
        for (; j < tp; ++j) dst[j] = (uint8_t)(dst[j] + p[j]);
    }
}

/* ---- similarityLocal (LL.cpp:1366-1428): 16x16 patch, row stride W/T ----------------------------- */
static void similarity_local16(const uint8_t *lm, const int32_t *feat, int nfeat, uint16_t *dst /*256*/,
                               int W, int H, int T, int cx, int cy)
{
    int Wd = W / T, Hd = H / T;
    memset(dst, 0, 256 * sizeof(uint16_t));
    int off_x = (cx / T - 8) * T, off_y = (cy / T - 8) * T;   /* LL.cpp:1380-1381 */
    const __m128i zero = _mm_setzero_si128();
    for (int i = 0; i < nfeat; ++i) {
        int fx = feat[3 * i] + off_x, fy = feat[3 * i + 1] + off_y, lab = feat[3 * i + 2];
        if (fx < 0 || fy < 0 || fx >= W || fy >= H) continue;   /* LL.cpp:1394 */
        const uint8_t *p = access_lm(lm, fx, fy, lab, T, Wd, Hd);
        for (int row = 0; row < 16; ++row) {
            __m128i lo = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)p), zero);
            __m128i hi = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)(p + 8)), zero);
            __m128i *d = (__m128i *)(dst + 16 * row);
            _mm_storeu_si128(d, _mm_add_epi16(_mm_loadu_si128(d), lo));
            _mm_storeu_si128(d + 1, _mm_add_epi16(_mm_loadu_si128(d + 1), hi));
            p += Wd;
        }
    }
}

/* ---- similarityLocal_64 (LL.cpp:1546-1620) ---------------------------------------------------------- */
static void similarity_local8(const uint8_t *lm, const int32_t *feat, int nfeat, uint8_t *dst /*256*/,
                              int W, int H, int T, int cx, int cy)
{
    int Wd = W / T, Hd = H / T;
    memset(dst, 0, 256);
    int off_x = (cx / T - 8) * T, off_y = (cy / T - 8) * T;
    for (int i = 0; i < nfeat; ++i) {
        int fx = feat[3 * i] + off_x, fy = feat[3 * i + 1] + off_y, lab = feat[3 * i + 2];
        if (fx < 0 || fy < 0 || fx >= W || fy >= H) continue;
        const uint8_t *p = access_lm(lm, fx, fy, lab, T, Wd, Hd);
        for (int row = 0; row < 16; ++row) {
            __m128i *d = (__m128i *)(dst + 16 * row);
            _mm_storeu_si128(d, _mm_add_epi8(_mm_loadu_si128(d), _mm_loadu_si128((const __m128i *)p)));
            p += Wd;
        }
    }
}

/* ---- matchClass (LL.cpp:1788-1941) --------------------------------------------------------------- */
typedef struct {
    int p0, p1, L;
    const int32_t *feat, *toff, *twh;
    const uint8_t *const *lm;
    const int *Ws, *Hs, *Ts;
    float thr;
    mo_match *out; long cap; long n;
    long coarse_cands, local_evals;
    int err;
} job_t;

static void push(job_t *jb, int x, int y, float s, int tid)
{
    if (jb->n < jb->cap) { mo_match *m = &jb->out[jb->n]; m->x = x; m->y = y; m->sim = s; m->cls = 0; m->tid = tid; }
    jb->n++;
}

static void *run_job(void *arg)
{
    job_t *jb = (job_t *)arg;
    int L = jb->L;
    int Wt = jb->Ws[L - 1], Ht = jb->Hs[L - 1], Tt = jb->Ts[L - 1];
    int Wd = Wt / Tt, Hd = Ht / Tt;
    size_t npos = (size_t)Wd * Hd;
    uint16_t *sim16 = (uint16_t *)malloc(npos * 2 * sizeof(uint16_t));
    uint8_t *sim8 = (uint8_t *)malloc(npos * 2);
    uint16_t *tot = (uint16_t *)malloc(npos * sizeof(uint16_t));
    size_t ccap = 1024, cn;
    mo_match *cand = (mo_match *)malloc(ccap * sizeof(mo_match));
    for (int p = jb->p0; p < jb->p1; ++p) {
        const int32_t *toff = jb->toff + (size_t)p * L * 2;
        const int32_t *twh = jb->twh + (size_t)p * L * 2 * 2;
        int lowest = (L - 1) * 2;
        int nf_total = 0, mode = -1;
        for (int m = 0; m < 2; ++m) {
            int e = lowest + m;
            int nf = toff[e + 1] - toff[e];
            nf_total += nf;
            if (nf > 8191) { jb->err = 1; goto done; }     /* CV_Assert LL.cpp:1291 (see header note) */
            if (mode <= 0) {                               /* LL.cpp:1813-1819 */
                if (nf < 64) mode = 1; else mode = 2;
            }
            const int32_t *f = jb->feat + 3 * (size_t)toff[e];
            if (mode == 1) {
                if (nf > 63) { jb->err = 1; goto done; }   /* CV_Assert LL.cpp:1457 */
                similarity8(jb->lm[lowest + m], f, nf, twh[2 * e], twh[2 * e + 1], sim8 + m * npos, Wt, Ht, Tt);
            } else if (mode == 2) {
                if (nf > 8191) { jb->err = 1; goto done; } /* CV_Assert LL.cpp:1291 */
                similarity16(jb->lm[lowest + m], f, nf, twh[2 * e], twh[2 * e + 1], sim16 + m * npos, Wt, Ht, Tt);
            }
        }
        if (mode == 1) for (size_t j = 0; j < npos; ++j) tot[j] = (uint16_t)(sim8[j] + sim8[npos + j]);
        else if (mode == 2) for (size_t j = 0; j < npos; ++j) tot[j] = (uint16_t)(sim16[j] + sim16[npos + j]);
        else memset(tot, 0, npos * sizeof(uint16_t));
        cn = 0;
        for (int r = 0; r < Hd; ++r)
            for (int c = 0; c < Wd; ++c) {
                int raw = tot[(size_t)r * Wd + c];
                float score = (raw * 100.f) / (4 * nf_total);      /* LL.cpp:1842 */
                if (score > jb->thr) {
                    int off = Tt / 2 + (Tt % 2 - 1);
                    if (cn == ccap) { ccap *= 2; cand = (mo_match *)realloc(cand, ccap * sizeof(mo_match)); }
                    cand[cn].x = c * Tt + off; cand[cn].y = r * Tt + off; cand[cn].sim = score; cand[cn].tid = p;
                    cn++;
                }
            }
        jb->coarse_cands += (long)cn;
        for (int l = L - 2; l >= 0; --l) {                          /* LL.cpp:1855-1938 */
            int T = jb->Ts[l], W = jb->Ws[l], H = jb->Hs[l];
            int start = l * 2;
            int border = 8 * T, offset = T / 2 + (T % 2 - 1);
            int max_x = W - twh[2 * start] - border, max_y = H - twh[2 * start + 1] - border;
            size_t kept = 0;
            for (size_t ci = 0; ci < cn; ++ci) {
                int x = cand[ci].x * 2 + 1, y = cand[ci].y * 2 + 1;
                if (x < border) x = border;
                if (y < border) y = border;
                if (x > max_x) x = max_x;
                if (y > max_y) y = max_y;
                int nfl = 0, md = -1;
                uint16_t l16[2][256];
                uint8_t l8[2][256];
                for (int m = 0; m < 2; ++m) {
                    int e = start + m;
                    int nf = toff[e + 1] - toff[e];
                    nfl += nf;
                    if (nf > 8191) { jb->err = 1; goto done; }
                    if (md <= 0) { if (nf < 64) md = 1; else md = 2; }
                    const int32_t *f = jb->feat + 3 * (size_t)toff[e];
                    if (md == 1) {
                        if (nf > 63) { jb->err = 1; goto done; }
                        similarity_local8(jb->lm[start + m], f, nf, l8[m], W, H, T, x, y);
                    } else if (md == 2) {
                        if (nf > 8191) { jb->err = 1; goto done; }
                        similarity_local16(jb->lm[start + m], f, nf, l16[m], W, H, T, x, y);
                    }
                }
                jb->local_evals++;
                float best = 0.f; int br = -1, bc = -1;
                for (int r = 0; r < 16; ++r)
                    for (int c = 0; c < 16; ++c) {
                        int raw = md == 1 ? l8[0][r * 16 + c] + l8[1][r * 16 + c]
                                : md == 2 ? (uint16_t)(l16[0][r * 16 + c] + l16[1][r * 16 + c]) : 0;
                        float score = (raw * 100.f) / (4 * nfl);    /* LL.cpp:1918 */
                        if (score > best) { best = score; br = r; bc = c; }
                    }
                cand[ci].sim = best;
                cand[ci].x = (x / T - 8 + bc) * T + offset;        /* LL.cpp:1930-1931 */
                cand[ci].y = (y / T - 8 + br) * T + offset;
                if (!(cand[ci].sim < jb->thr)) cand[kept++] = cand[ci];   /* remove_if(sim < thr) :1935 */
            }
            cn = kept;
        }
        for (size_t ci = 0; ci < cn; ++ci) push(jb, cand[ci].x, cand[ci].y, cand[ci].sim, p);
    }
done:
    free(sim16); free(sim8); free(tot); free(cand);
    return NULL;
}

/*
 * One class.  feat: (F,3) int32; toff: (P*L*2+1) offsets; twh: (P*L*2,2) width,height;
 * lm[l*2+m]: linear-memory block; returns the number of matches (may exceed cap: call again),
 * or -1 on a reference CV_Assert.  stats[0]=coarse candidates, stats[1]=local 16x16 evaluations.
 * nthreads>1 splits template pyramids across pthreads (NOT what the reference does; a labelled
 * variant for the baseline report) — output order then differs, the multiset does not.
 */
long mo_match_bank(int P, int L, const int32_t *feat, const int32_t *toff, const int32_t *twh,
                   const uint8_t *const *lm, const int *Ws, const int *Hs, const int *Ts, float thr,
                   mo_match *out, long cap, int nthreads, long *stats)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t jobs[256];
    pthread_t th[256];
    long total = 0;
    /* pass 1 runs the jobs into private buffers sized cap each would be wasteful; instead run
       each job with a private heap buffer that grows, then concatenate. */
    for (int t = 0; t < nthreads; ++t) {
        job_t *jb = &jobs[t];
        memset(jb, 0, sizeof(*jb));
        jb->p0 = (int)((long)P * t / nthreads);
        jb->p1 = (int)((long)P * (t + 1) / nthreads);
        jb->L = L; jb->feat = feat; jb->toff = toff; jb->twh = twh; jb->lm = lm;
        jb->Ws = Ws; jb->Hs = Hs; jb->Ts = Ts; jb->thr = thr;
        jb->cap = cap; jb->out = (mo_match *)malloc(sizeof(mo_match) * (size_t)(cap > 0 ? cap : 1));
    }
    if (nthreads == 1) run_job(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, run_job, &jobs[t]);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    int err = 0;
    stats[0] = stats[1] = 0;
    for (int t = 0; t < nthreads; ++t) {
        job_t *jb = &jobs[t];
        err |= jb->err;
        long w = jb->n < jb->cap ? jb->n : jb->cap;
        for (long i = 0; i < w && total + i < cap; ++i) out[total + i] = jb->out[i];
        total += jb->n;
        stats[0] += jb->coarse_cands; stats[1] += jb->local_evals;
        free(jb->out);
    }
    return err ? -1 : total;
}
