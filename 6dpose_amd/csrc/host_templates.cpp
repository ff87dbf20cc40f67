// Host-side template extraction and YAML I/O (see host_templates.h).  Follows the behaviour of
// LL.cpp:234-318 (crop / scatter), :589-643 (colour extraction), :888-966 (normal extraction),
// :2043-2122 (class YAML) and the OpenCV op semantics of SURVEY Appendix A.8-10.
#include "host_templates.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <limits>
#include <sstream>

namespace lm {

namespace {

struct Candidate {            // LL.h:79-91
    Feature f;
    float score;
};

// getLabel (LL.cpp:176-192); -1 when not one-hot
inline int get_label(int q) {
    if (q <= 0 || (q & (q - 1)) || q > 128) return -1;
    int l = 0;
    while ((1 << l) != q) ++l;
    return l;
}

// cv::erode(3x3 rect, BORDER_REPLICATE) = 3x3 minimum (LL.cpp:595, 894)
std::vector<uint8_t> erode3(const std::vector<uint8_t>& src, int W, int H) {
    std::vector<uint8_t> dst(src.size());
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t m = 255;
            for (int dy = -1; dy <= 1; ++dy) {
                int yy = std::min(std::max(y + dy, 0), H - 1);
                for (int dx = -1; dx <= 1; ++dx) {
                    int xx = std::min(std::max(x + dx, 0), W - 1);
                    m = std::min(m, src[(size_t)yy * W + xx]);
                }
            }
            dst[(size_t)y * W + x] = m;
        }
    return dst;
}

// cv::distanceTransform(src, DIST_C, 3) (LL.cpp:905): chessboard distance to the nearest zero pixel,
// exact with the two-pass 3x3 chamfer (all weights 1); outside the image counts as "far".
std::vector<float> chessboard_dt(const std::vector<uint8_t>& nz, int W, int H) {
    const int INF = 1 << 28;
    std::vector<int> d((size_t)W * H);
    for (size_t i = 0; i < d.size(); ++i) d[i] = nz[i] ? INF : 0;
    auto at = [&](int y, int x) -> int { return (y < 0 || x < 0 || y >= H || x >= W) ? INF : d[(size_t)y * W + x]; };
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int v = d[(size_t)y * W + x];
            if (!v) continue;
            int m = std::min(std::min(at(y - 1, x - 1), at(y - 1, x)), std::min(at(y - 1, x + 1), at(y, x - 1)));
            d[(size_t)y * W + x] = std::min(v, m + 1);
        }
    for (int y = H - 1; y >= 0; --y)
        for (int x = W - 1; x >= 0; --x) {
            int v = d[(size_t)y * W + x];
            if (!v) continue;
            int m = std::min(std::min(at(y + 1, x + 1), at(y + 1, x)), std::min(at(y + 1, x - 1), at(y, x + 1)));
            d[(size_t)y * W + x] = std::min(v, m + 1);
        }
    std::vector<float> out(d.size());
    for (size_t i = 0; i < d.size(); ++i) out[i] = (float)d[i];
    return out;
}

// QuantizedPyramid::selectScatteredFeatures (LL.cpp:279-318)
bool select_scattered(const std::vector<Candidate>& cands, std::vector<Feature>& features, size_t num_features,
                      float distance) {
    features.clear();
    if (cands.empty()) return false;
    float distance_sq = distance * distance;
    int i = 0;
    long guard = 0;
    while (features.size() < num_features) {
        const Candidate& c = cands[i];
        bool keep = true;
        for (size_t j = 0; j < features.size() && keep; ++j) {
            const Feature& f = features[j];
            keep = (float)((c.f.x - f.x) * (c.f.x - f.x) + (c.f.y - f.y) * (c.f.y - f.y)) >= distance_sq;
        }
        if (keep) features.push_back(c.f);
        if (++i == (int)cands.size()) {
            i = 0;
            distance -= 1.0f;
            distance_sq = distance * distance;
            if (++guard > 1000000) return false;   // duplicate candidate positions: cannot terminate
        }
    }
    return features.size() == num_features;
}

struct ScoreDesc {
    bool operator()(const Candidate& a, const Candidate& b) const { return a.score > b.score; }   // LL.h:84-87
};

}  // namespace

static bool extract_color_full(const float* mag, const uint8_t* angle, const uint8_t* mask, int W, int H,
                               size_t num_features, float strong_threshold, int level, Template& out) {
    std::vector<uint8_t> local;
    if (mask) {
        std::vector<uint8_t> m(mask, mask + (size_t)W * H);
        std::vector<uint8_t> er = erode3(m, W, H);
        local.resize(m.size());
        for (size_t i = 0; i < m.size(); ++i) local[i] = (uint8_t)std::max(0, (int)m[i] - (int)er[i]);   // cv::subtract
    }
    std::vector<Candidate> cands;
    const float thr_sq = strong_threshold * strong_threshold;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            size_t o = (size_t)r * W + c;
            if (mask && !local[o]) continue;
            uint8_t q = angle[o];
            if (q > 0 && mag[o] > thr_sq) {
                int lab = get_label(q);
                if (lab < 0) return false;
                cands.push_back(Candidate{Feature{c, r, lab}, mag[o]});
            }
        }
    if (cands.size() < num_features) return false;                      // LL.cpp:626
    std::stable_sort(cands.begin(), cands.end(), ScoreDesc());        // LL.cpp:629
    float distance = static_cast<float>(cands.size() / num_features + 1);   // LL.cpp:632
    if (!select_scattered(cands, out.features, num_features, distance)) return false;
    out.width = out.height = -1;
    out.pyramid_level = level;
    return true;
}

static bool extract_normal_full(const uint8_t* normal, const uint8_t* mask, int W, int H, size_t num_features,
                                int extract_threshold, int level, Template& out) {
    const size_t N = (size_t)W * H;
    std::vector<uint8_t> local;
    if (mask) {
        std::vector<uint8_t> m(mask, mask + N);
        local = erode3(erode3(m, W, H), W, H);                           // iterations = 2, LL.cpp:894
    }
    std::vector<float> dist[8];
    std::vector<uint8_t> temp(N);
    for (int i = 0; i < 8; ++i) {
        for (size_t o = 0; o < N; ++o) temp[o] = ((!mask || local[o]) && (normal[o] & (1 << i))) ? 1 : 0;
        dist[i] = chessboard_dt(temp, W, H);
    }
    int label_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<Candidate> cands;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            size_t o = (size_t)r * W + c;
            if (mask && !local[o]) continue;
            uint8_t q = normal[o];
            if (q != 0 && q != 255) {
                int lab = get_label(q);
                if (lab < 0) return false;
                float score = dist[lab][o];
                if (score >= (float)extract_threshold) {
                    cands.push_back(Candidate{Feature{c, r, lab}, score});
                    ++label_counts[lab];
                }
            }
        }
    if (cands.size() < num_features) return false;                      // LL.cpp:943
    for (Candidate& c : cands) c.score /= (float)label_counts[c.f.label];
    std::stable_sort(cands.begin(), cands.end(), ScoreDesc());
    float area = 0.f;
    if (!mask) area = (float)N;
    else { size_t nz = 0; for (size_t o = 0; o < N; ++o) nz += local[o] != 0; area = (float)nz; }
    float distance = sqrtf(area) / sqrtf((float)num_features) + 1.5f;    // LL.cpp:957
    select_scattered(cands, out.features, num_features, distance);      // return value ignored, LL.cpp:958
    out.width = out.height = -1;
    out.pyramid_level = level;
    return true;
}

// With an object mask every candidate lies inside it, and erosion / the chessboard distance transform at a masked pixel
// only depend on pixels up to the first zero around it.  The mask's bounding box grown by one pixel (all zeros, or the
// image border itself, where the full-image code clamps in the same way) therefore gives exactly the same templates
// as the whole frame, at the cost of the object's pixels instead of the image's (render_train: ~30x less host work).
namespace {
struct Roi { int x0, y0, w, h; bool any; };
Roi mask_roi(const uint8_t* mask, int W, int H) {
    int x0 = W, y0 = H, x1 = -1, y1 = -1;
    for (int y = 0; y < H; ++y) {
        const uint8_t* row = mask + (size_t)y * W;
        int first = -1, last = -1;
        for (int x = 0; x < W; ++x)
            if (row[x]) { if (first < 0) first = x; last = x; }
        if (first >= 0) { x0 = std::min(x0, first); x1 = std::max(x1, last); y0 = std::min(y0, y); y1 = y; }
    }
    Roi r{0, 0, 0, 0, x1 >= 0};
    if (!r.any) return r;
    x0 = std::max(x0 - 1, 0); y0 = std::max(y0 - 1, 0); x1 = std::min(x1 + 1, W - 1); y1 = std::min(y1 + 1, H - 1);
    r.x0 = x0; r.y0 = y0; r.w = x1 - x0 + 1; r.h = y1 - y0 + 1;
    return r;
}
template <typename T>
std::vector<T> crop(const T* src, int W, const Roi& r) {
    std::vector<T> out((size_t)r.w * r.h);
    for (int y = 0; y < r.h; ++y) memcpy(&out[(size_t)y * r.w], src + (size_t)(r.y0 + y) * W + r.x0, (size_t)r.w * sizeof(T));
    return out;
}
}  // namespace

bool extract_color_template(const float* mag, const uint8_t* angle, const uint8_t* mask, int W, int H,
                            size_t num_features, float strong_threshold, int level, Template& out) {
    if (!mask) return extract_color_full(mag, angle, nullptr, W, H, num_features, strong_threshold, level, out);
    const Roi r = mask_roi(mask, W, H);
    if (!r.any) return false;                                            // no candidates at all (LL.cpp:626)
    const std::vector<float> m = crop(mag, W, r);
    const std::vector<uint8_t> a = crop(angle, W, r), k = crop(mask, W, r);
    if (!extract_color_full(m.data(), a.data(), k.data(), r.w, r.h, num_features, strong_threshold, level, out)) return false;
    for (Feature& f : out.features) { f.x += r.x0; f.y += r.y0; }
    return true;
}

bool extract_normal_template(const uint8_t* normal, const uint8_t* mask, int W, int H, size_t num_features,
                             int extract_threshold, int level, Template& out) {
    if (!mask) return extract_normal_full(normal, nullptr, W, H, num_features, extract_threshold, level, out);
    const Roi r = mask_roi(mask, W, H);
    if (!r.any) return false;                                            // no candidates at all (LL.cpp:943)
    const std::vector<uint8_t> n = crop(normal, W, r), k = crop(mask, W, r);
    if (!extract_normal_full(n.data(), k.data(), r.w, r.h, num_features, extract_threshold, level, out)) return false;
    for (Feature& f : out.features) { f.x += r.x0; f.y += r.y0; }
    return true;
}

void crop_templates(TemplatePyramid& tp) {
    int min_x = std::numeric_limits<int>::max(), min_y = min_x;
    int max_x = std::numeric_limits<int>::min(), max_y = max_x;
    for (const Template& t : tp)
        for (const Feature& f : t.features) {
            int x = f.x << t.pyramid_level, y = f.y << t.pyramid_level;
            min_x = std::min(min_x, x); min_y = std::min(min_y, y);
            max_x = std::max(max_x, x); max_y = std::max(max_y, y);
        }
    if (min_x % 2 == 1) --min_x;
    if (min_y % 2 == 1) --min_y;
    for (Template& t : tp) {
        t.width = (max_x - min_x) >> t.pyramid_level;
        t.height = (max_y - min_y) >> t.pyramid_level;
        int ox = min_x >> t.pyramid_level, oy = min_y >> t.pyramid_level;
        for (Feature& f : t.features) { f.x -= ox; f.y -= oy; }
    }
}

// ---- YAML -------------------------------------------------------------------------------------

bool write_class_yaml(const std::string& path, const std::string& class_id, const std::vector<TemplatePyramid>& tps,
                      int pyramid_levels, std::string& err) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) { err = "cannot open for writing: " + path; return false; }
    fprintf(f, "%%YAML:1.0\n---\nclass_id: \"%s\"\nmodalities: [ ColorGradient, DepthNormal ]\n", class_id.c_str());
    fprintf(f, "pyramid_levels: %d\ntemplate_pyramids:\n", pyramid_levels);
    for (size_t i = 0; i < tps.size(); ++i) {
        fprintf(f, "   -\n      template_id: %d\n      templates:\n", (int)i);
        for (const Template& t : tps[i]) {
            fprintf(f, "         -\n            width: %d\n            height: %d\n            pyramid_level: %d\n",
                    t.width, t.height, t.pyramid_level);
            fprintf(f, "            features:\n");
            for (const Feature& ft : t.features) fprintf(f, "               - [ %d, %d, %d ]\n", ft.x, ft.y, ft.label);
        }
    }
    bool ok = fclose(f) == 0;
    if (!ok) err = "write failed: " + path;
    return ok;
}

namespace {
inline std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
inline bool starts(const std::string& s, const char* p) { return s.compare(0, strlen(p), p) == 0; }
}  // namespace

bool read_class_yaml(const std::string& path, std::string& class_id, std::vector<std::string>& modalities,
                     int& pyramid_levels, std::vector<TemplatePyramid>& tps, std::string& err) {
    std::ifstream in(path);
    if (!in) { err = "cannot open: " + path; return false; }
    class_id.clear(); modalities.clear(); tps.clear();
    pyramid_levels = -1;
    std::string line;
    Template* cur = nullptr;
    int expected = 0;
    while (std::getline(in, line)) {
        std::string s = trim(line);
        if (s.empty() || s[0] == '%' || s == "---") continue;
        if (starts(s, "- [") || starts(s, "-[")) {
            if (!cur) { err = "feature outside a template"; return false; }
            int x, y, l;
            if (sscanf(s.c_str() + 1, " [ %d , %d , %d ]", &x, &y, &l) != 3) { err = "bad feature line: " + s; return false; }
            cur->features.push_back(Feature{x, y, l});
        } else if (starts(s, "class_id:")) {
            std::string v = trim(s.substr(9));
            if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
            class_id = v;
        } else if (starts(s, "modalities:")) {
            size_t a = s.find('['), b = s.rfind(']');
            if (a == std::string::npos || b == std::string::npos) { err = "bad modalities line"; return false; }
            std::stringstream ss(s.substr(a + 1, b - a - 1));
            std::string tok;
            while (std::getline(ss, tok, ',')) { tok = trim(tok); if (!tok.empty()) modalities.push_back(tok); }
        } else if (starts(s, "pyramid_levels:")) {
            pyramid_levels = atoi(s.c_str() + 15);
        } else if (starts(s, "template_id:")) {
            int tid = atoi(s.c_str() + 12);
            if (tid != expected) { err = "template_id == expected_id (LL.cpp:2077)"; return false; }
            ++expected;
            tps.emplace_back();
            cur = nullptr;
        } else if (starts(s, "width:")) {
            if (tps.empty()) { err = "template outside a pyramid"; return false; }
            tps.back().emplace_back();
            cur = &tps.back().back();
            cur->width = atoi(s.c_str() + 6);
        } else if (starts(s, "height:")) {
            if (cur) cur->height = atoi(s.c_str() + 7);
        } else if (starts(s, "pyramid_level:")) {
            if (cur) cur->pyramid_level = atoi(s.c_str() + 14);
        }
        // other keys (e.g. the obsolete `depth:` of older banks) are ignored like cv::FileNode lookups do
    }
    if (pyramid_levels < 0) { err = "no pyramid_levels in " + path; return false; }
    return true;
}

}  // namespace lm
