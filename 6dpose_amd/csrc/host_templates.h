// Host-side template bank: structures, greedy feature extraction (Detector::addTemplate back half)
// and the OpenCV-FileStorage YAML subset used by writeClass/readClass.  Sequential by nature
// (stable sort + greedy scatter), so it stays on the host; the quantised maps it
// consumes come from the HIP front end.
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

namespace lm {

struct Feature { int x, y, label; };                       // LL.h:23-34

struct Template {                                           // LL.h:36-45
    int width = -1, height = -1, pyramid_level = 0;
    std::vector<Feature> features;
};
using TemplatePyramid = std::vector<Template>;              // LL.h:361: [l0 colour, l0 normal, l1 colour, ...]
using TemplatesMap = std::map<std::string, std::vector<TemplatePyramid>>;   // LL.h:362

// ColorGradientPyramid::extractTemplate (LL.cpp:589-643).  mask may be null (no mask).
bool extract_color_template(const float* mag, const uint8_t* angle, const uint8_t* mask, int W, int H,
                            size_t num_features, float strong_threshold, int level, Template& out);
// DepthNormalPyramid::extractTemplate (LL.cpp:888-966).
bool extract_normal_template(const uint8_t* normal, const uint8_t* mask, int W, int H, size_t num_features,
                             int extract_threshold, int level, Template& out);
// cropTemplates (LL.cpp:234-277)
void crop_templates(TemplatePyramid& tp);

// writeClass / readClass (LL.cpp:2043-2122).  Return false and fill err on failure.
bool write_class_yaml(const std::string& path, const std::string& class_id, const std::vector<TemplatePyramid>& tps,
                      int pyramid_levels, std::string& err);
bool read_class_yaml(const std::string& path, std::string& class_id, std::vector<std::string>& modalities,
                     int& pyramid_levels, std::vector<TemplatePyramid>& tps, std::string& err);

}  // namespace lm
