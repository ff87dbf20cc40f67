/*
 * TEST INFRASTRUCTURE (oracle/_ref), NOT product code.
 *
 * Runs the reference's OWN match code: this translation unit includes
 *   - /root/reference/linemodLevelup/linemodLevelup.h            (directly, via -I; never copied)
 *   - oracle/_ref/ll_match_extract.inc = lines 1022-1658 and 1694-1941 of
 *     /root/reference/linemodLevelup/linemodLevelup.cpp, cut out by oracle/Makefile at build time
 *     into the git-ignored oracle/_ref/ (orUnaligned8u, spread, SIMILARITY_LUT,
 *     computeResponseMaps, linearize, accessLinearMemory, similarity, similarityLocal,
 *     addSimilarities, the _64 variants, Detector(modalities, T), Detector::match,
 *     MatchPredicate, Detector::matchClass)
 * against the buffer-type shim in oracle/ref_shim/opencv2/.  The quantised 8-bit maps (the
 * output of the OpenCV-dependent half, LL.cpp:350-880, which cannot be built here and which the
 * reference's golden YAML pins in oracle/linemod_oracle.py) are handed in by the caller through a
 * Modality whose QuantizedPyramid returns them level by level; from there on every instruction
 * executed is the reference's.
 *
 * Only tests/ and tests/golden/make_ref_fixtures.py load the resulting library.
 */
#include "linemodLevelup.h"

#include <cstdint>
#include <cstdio>

using namespace std;
using namespace cv;

namespace linemodLevelup {
#include "ll_match_extract.inc"
}  // namespace linemodLevelup

namespace {
using namespace linemodLevelup;

/* The caller's quantised maps, one per pyramid level, behind the reference's modality seam
 * (LL.h:50-148): quantize() hands out the current level, pyrDown() steps to the next. */
class GivenPyramid : public QuantizedPyramid {
public:
    explicit GivenPyramid(const std::vector<Mat>& l) : levels_(l), cur_(0) {}
    void quantize(Mat& dst) const override { dst = levels_[cur_].clone(); }
    bool extractTemplate(Template&) const override { return false; }
    void pyrDown() override { ++cur_; }
private:
    std::vector<Mat> levels_;
    size_t cur_;
};

class GivenModality : public Modality {
public:
    explicit GivenModality(const std::vector<Mat>& l) : levels_(l) {}
    std::string name() const override { return "Given"; }
    void read(const FileNode&) override {}
    void write(FileStorage&) const override {}
protected:
    Ptr<QuantizedPyramid> processImpl(const std::vector<Mat>&, const Mat&) const override
    {
        return makePtr<GivenPyramid>(levels_);
    }
private:
    std::vector<Mat> levels_;
};

/* Access to the protected bank and to matchClass. */
class RefDetector : public Detector {
public:
    RefDetector(const std::vector<Ptr<Modality> >& m, const std::vector<int>& T) : Detector(m, T) {}
    std::vector<TemplatePyramid>& bank(const std::string& cls) { return class_templates[cls]; }
    bool has(const std::string& cls) const { return class_templates.find(cls) != class_templates.end(); }

    /* The front half of Detector::match (LL.cpp:1721-1752) re-enacted with the reference's own
     * static functions, then matchClass per class WITHOUT the final sort/unique, so that the
     * pre-unique list (order-free parity) can be read. */
    std::vector<Match> matchPreUnique(const std::vector<std::vector<Mat> >& quantized /*[modality][level]*/,
                                      float threshold, const std::vector<std::string>& class_ids) const
    {
        LinearMemoryPyramid lm_pyramid(pyramid_levels, std::vector<LinearMemories>(modalities.size(), LinearMemories(8)));
        std::vector<Size> sizes;
        for (int l = 0; l < pyramid_levels; ++l) {
            int T = T_at_level[l];
            Mat spread_quantized;
            std::vector<Mat> response_maps;
            for (size_t i = 0; i < modalities.size(); ++i) {
                spread(quantized[i][l], spread_quantized, T);
                computeResponseMaps(spread_quantized, response_maps);
                for (int j = 0; j < 8; ++j) linearize(response_maps[j], lm_pyramid[l][i][j], T);
            }
            sizes.push_back(quantized.back()[l].size());
        }
        std::vector<Match> matches;
        if (class_ids.empty()) {
            for (TemplatesMap::const_iterator it = class_templates.begin(); it != class_templates.end(); ++it)
                matchClass(lm_pyramid, sizes, threshold, matches, it->first, it->second);
        } else {
            for (size_t i = 0; i < class_ids.size(); ++i) {
                TemplatesMap::const_iterator it = class_templates.find(class_ids[i]);
                if (it != class_templates.end()) matchClass(lm_pyramid, sizes, threshold, matches, it->first, it->second);
            }
        }
        return matches;
    }
};

thread_local std::string g_err;
}  // namespace

extern "C" {

struct ref_match { int32_t x, y; float sim; int32_t cls, tid; };

const char* ref_last_error() { return g_err.c_str(); }

/* Which SIMD paths this build of the reference lines takes (CV_SSE2 | CV_SSE3<<1 | CV_SSSE3<<2). */
int ref_simd_flags()
{
    int f = 0;
#if CV_SSE2
    f |= 1;
#endif
#if CV_SSE3
    f |= 2;
#endif
#if CV_SSSE3
    f |= 4;
#endif
    return f;
}

/* spread (LL.cpp:1094-1109) of one quantised map. */
int ref_spread(const uint8_t* q, int W, int H, int T, uint8_t* out)
{
    try {
        Mat src(H, W, CV_8U, const_cast<uint8_t*>(q)), dst;
        Mat owned = src.clone();
        spread(owned, dst, T);
        for (int r = 0; r < H; ++r) memcpy(out + size_t(r) * W, dst.ptr(r), W);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

/* spread -> computeResponseMaps -> linearize (LL.cpp:1094-1243) of one quantised map.
 * out = u8[8 labels][T*T phases][(W/T)*(H/T)], no tail. */
int ref_build_linear_memories(const uint8_t* q, int W, int H, int T, uint8_t* out)
{
    try {
        Mat src = Mat(H, W, CV_8U, const_cast<uint8_t*>(q)).clone(), spr, lin;
        std::vector<Mat> resp;
        spread(src, spr, T);
        computeResponseMaps(spr, resp);
        size_t per = size_t(W) * H;
        for (int j = 0; j < 8; ++j) {
            linearize(resp[j], lin, T);
            for (int p = 0; p < T * T; ++p) memcpy(out + j * per + size_t(p) * lin.cols, lin.ptr(p), lin.cols);
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

/*
 * Detector::match (mode 0: LL.cpp:1702-1777 verbatim, i.e. std::sort + std::unique applied) or the
 * pre-unique concatenation of matchClass (mode 1).
 *
 * quantized[level*2 + modality] = u8 maps of Ws[level] x Hs[level].
 * Bank: classes in `class_names` order, class c owns pyramids [pyr_start[c], pyr_start[c+1]);
 * pyramid p has levels*2 templates (level-major, modality-minor: LL.h:336-337), template k covers
 * feat[tmpl_off[k] .. tmpl_off[k+1]) rows of (x, y, label) and has tmpl_wh[k] = (width, height).
 * req: the class_ids argument of match (n_req == 0 -> all classes in std::map order).
 * out[i].cls = index into class_names.  Returns the number of matches (may exceed cap; only cap are
 * written), or -1 on a cv::Exception (message in ref_last_error()).
 */
long ref_match(int levels, const int* T, const int* Ws, const int* Hs, const uint8_t* const* quantized,
               int num_classes, const char* const* class_names, const int* pyr_start,
               const int32_t* feat, const int32_t* tmpl_off, const int32_t* tmpl_wh,
               float threshold, int n_req, const char* const* req, int mode, ref_match* out, long cap)
{
    try {
        std::vector<std::vector<Mat> > q(2, std::vector<Mat>(levels));
        for (int l = 0; l < levels; ++l)
            for (int m = 0; m < 2; ++m)
                q[m][l] = Mat(Hs[l], Ws[l], CV_8U, const_cast<uint8_t*>(quantized[l * 2 + m])).clone();
        std::vector<Ptr<Modality> > mods;
        mods.push_back(makePtr<GivenModality>(q[0]));
        mods.push_back(makePtr<GivenModality>(q[1]));
        RefDetector det(mods, std::vector<int>(T, T + levels));
        std::map<std::string, int> cls_index;
        for (int c = 0; c < num_classes; ++c) {
            cls_index[class_names[c]] = c;
            std::vector<std::vector<Template> >& bank = det.bank(class_names[c]);
            for (int p = pyr_start[c]; p < pyr_start[c + 1]; ++p) {
                std::vector<Template> tp(size_t(levels) * 2);
                for (int k = 0; k < levels * 2; ++k) {
                    int e = p * levels * 2 + k;
                    Template& t = tp[k];
                    t.width = tmpl_wh[2 * e];
                    t.height = tmpl_wh[2 * e + 1];
                    t.pyramid_level = k / 2;
                    for (int f = tmpl_off[e]; f < tmpl_off[e + 1]; ++f)
                        t.features.push_back(Feature(feat[3 * f], feat[3 * f + 1], feat[3 * f + 2]));
                }
                bank.push_back(tp);
            }
        }
        std::vector<std::string> ids;
        for (int i = 0; i < n_req; ++i) ids.push_back(req[i]);
        std::vector<Match> res;
        if (mode == 0) {
            std::vector<Mat> sources(2);      // only sources.size() and (empty) masks are looked at
            res = det.match(sources, threshold, ids);
        } else {
            res = det.matchPreUnique(q, threshold, ids);
        }
        long n = long(res.size());
        for (long i = 0; i < n && i < cap; ++i) {
            out[i].x = res[i].x; out[i].y = res[i].y; out[i].sim = res[i].similarity;
            out[i].cls = cls_index[res[i].class_id]; out[i].tid = res[i].template_id;
        }
        return n;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"
