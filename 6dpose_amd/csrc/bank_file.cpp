// Packed binary template bank.  The reference stores one OpenCV-YAML file per class
// (Detector::writeClasses / readClasses, LL.cpp:2124-2146; schema LL.cpp:2043-2122): ~45 text bytes per feature
// and a parser in front of it, which is what makes the 16k- and 90k-template configurations slow to load.  This file
// holds the same information — class id, per template width / height / pyramid level, per feature x, y, label — as
// flat arrays that are mmap'ed and walked once:
//
//   BankHeader                      magic, version, pyramid_levels, class count, directory offset, file size
//   per class:  TemplRec[P*E]       width, height, first feature (index into the class's feature array); E = levels*2,
//                                   reference TemplatePyramid order (LL.h:336-337), +1 end record
//               uint32 feat[n]      x:int16 | y:int13 << 16 | label << 29
//   directory:  ClassRec[classes]   name offset/length, pyramid count, offsets of the two arrays
//   names
//
// Every section starts on an 8-byte boundary.  Reading takes a class filter, so a rank that serves a subset of the
// objects (configs[3]: one object per GPU) touches only those pages of the mapping.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>

#include "detector_internal.h"

namespace {

constexpr char kMagic[8] = {'L', 'M', 'B', 'A', 'N', 'K', '0', '1'};
constexpr uint32_t kVersion = 1;

struct BankHeader {
    char magic[8];
    uint32_t version, pyramid_levels, num_classes, reserved;
    uint64_t dir_offset, file_bytes;
};
struct ClassRec {
    uint64_t name_offset;
    uint32_t name_len, num_pyramids;
    uint64_t templ_offset, feat_offset, num_features;
};
struct TemplRec {
    int32_t width, height;
    uint64_t feat_begin;
};
static_assert(sizeof(BankHeader) == 40 && sizeof(ClassRec) == 40 && sizeof(TemplRec) == 16, "bank file layout");

inline bool pack_feature(const Feature& f, uint32_t& out) {
    if (f.x < -32768 || f.x > 32767 || f.y < -4096 || f.y > 4095 || f.label < 0 || f.label > 7) return false;
    out = (uint32_t)(uint16_t)(int16_t)f.x | (((uint32_t)f.y & 0x1FFFu) << 16) | ((uint32_t)f.label << 29);
    return true;
}
inline Feature unpack_feature(uint32_t v) {
    Feature f;
    f.x = (int16_t)(v & 0xFFFFu);
    f.y = ((int32_t)((v >> 16) & 0x1FFFu) ^ 0x1000) - 0x1000;   // sign-extend 13 bits
    f.label = (int)(v >> 29);
    return f;
}

struct Writer {
    FILE* f = nullptr;
    uint64_t pos = 0;
    bool ok = true;
    void put(const void* p, size_t n) {
        if (n && fwrite(p, 1, n, f) != n) ok = false;
        pos += n;
    }
    void align8() {
        static const char z[8] = {0};
        if (pos & 7) put(z, 8 - (pos & 7));
    }
};

}  // namespace

extern "C" int lm_detector_write_bank(const lm_detector* d, const char* path, const char* const* class_ids, int num_class_ids) {
    if (!d || !path || num_class_ids < 0 || (num_class_ids && !class_ids)) return lm_set_error(LM_ERR_INVALID, "bad argument");
    std::vector<const std::pair<const std::string, std::vector<TemplatePyramid>>*> sel;
    if (num_class_ids == 0) {
        for (auto& kv : d->class_templates) sel.push_back(&kv);
    } else {
        for (int i = 0; i < num_class_ids; ++i) {
            auto it = d->class_templates.find(class_ids[i] ? class_ids[i] : "");
            if (it == d->class_templates.end()) return lm_set_error(LM_ERR_NOT_FOUND, "unknown class '%s'", class_ids[i] ? class_ids[i] : "");
            sel.push_back(&*it);
        }
    }
    Writer w;
    w.f = fopen(path, "wb");
    if (!w.f) return lm_set_error(LM_ERR_IO, "cannot open for writing: %s (%s)", path, strerror(errno));
    BankHeader h{};
    memcpy(h.magic, kMagic, 8);
    h.version = kVersion; h.pyramid_levels = (uint32_t)d->pyramid_levels; h.num_classes = (uint32_t)sel.size();
    w.put(&h, sizeof(h));
    std::vector<ClassRec> dir(sel.size());
    std::vector<TemplRec> recs;
    std::vector<uint32_t> feats;
    const size_t E = (size_t)d->pyramid_levels * 2;
    for (size_t c = 0; c < sel.size(); ++c) {
        const std::vector<TemplatePyramid>& tps = sel[c]->second;
        recs.clear(); feats.clear();
        recs.reserve(tps.size() * E + 1);
        for (const TemplatePyramid& tp : tps)
            for (const Template& t : tp) {
                recs.push_back(TemplRec{t.width, t.height, (uint64_t)feats.size()});
                for (const Feature& f : t.features) {
                    uint32_t v;
                    if (!pack_feature(f, v)) {
                        fclose(w.f); unlink(path);
                        return lm_set_error(LM_ERR_INVALID, "feature (%d,%d,%d) of class '%s' does not fit the packed format (x int16, y int13, label 0..7)",
                                            f.x, f.y, f.label, sel[c]->first.c_str());
                    }
                    feats.push_back(v);
                }
            }
        recs.push_back(TemplRec{0, 0, (uint64_t)feats.size()});
        ClassRec& cr = dir[c];
        cr.num_pyramids = (uint32_t)tps.size();
        cr.num_features = feats.size();
        w.align8(); cr.templ_offset = w.pos; w.put(recs.data(), recs.size() * sizeof(TemplRec));
        w.align8(); cr.feat_offset = w.pos; w.put(feats.data(), feats.size() * sizeof(uint32_t));
    }
    w.align8();
    uint64_t names = w.pos + dir.size() * sizeof(ClassRec);
    for (size_t c = 0; c < sel.size(); ++c) {
        dir[c].name_offset = names; dir[c].name_len = (uint32_t)sel[c]->first.size();
        names += sel[c]->first.size();
    }
    h.dir_offset = w.pos;
    w.put(dir.data(), dir.size() * sizeof(ClassRec));
    for (auto* kv : sel) w.put(kv->first.data(), kv->first.size());
    h.file_bytes = w.pos;
    if (fseek(w.f, 0, SEEK_SET) != 0 || fwrite(&h, 1, sizeof(h), w.f) != sizeof(h)) w.ok = false;
    if (fclose(w.f) != 0) w.ok = false;
    if (!w.ok) { unlink(path); return lm_set_error(LM_ERR_IO, "write failed: %s", path); }
    return LM_OK;
}

namespace {

struct Mapping {
    const uint8_t* p = nullptr;
    size_t n = 0;
    ~Mapping() { if (p) munmap((void*)p, n); }
    bool span(uint64_t off, uint64_t bytes) const { return off <= n && bytes <= n - off && (off & 7) == 0; }
};

int map_bank(const char* path, Mapping& m, BankHeader& h, const ClassRec*& dir) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return lm_set_error(LM_ERR_IO, "cannot open: %s (%s)", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < (off_t)sizeof(BankHeader)) { close(fd); return lm_set_error(LM_ERR_IO, "not a bank file: %s", path); }
    void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return lm_set_error(LM_ERR_IO, "mmap failed: %s (%s)", path, strerror(errno));
    m.p = (const uint8_t*)p; m.n = (size_t)st.st_size;
    memcpy(&h, m.p, sizeof(h));
    if (memcmp(h.magic, kMagic, 8) != 0 || h.version != kVersion) return lm_set_error(LM_ERR_IO, "not a bank file (magic / version): %s", path);
    if (h.file_bytes != m.n) return lm_set_error(LM_ERR_IO, "bank file truncated: header says %llu bytes, file has %zu", (unsigned long long)h.file_bytes, m.n);
    if (!m.span(h.dir_offset, (uint64_t)h.num_classes * sizeof(ClassRec))) return lm_set_error(LM_ERR_IO, "bank directory outside the file");
    dir = (const ClassRec*)(m.p + h.dir_offset);
    for (uint32_t c = 0; c < h.num_classes; ++c)
        if (dir[c].name_offset > m.n || dir[c].name_len > m.n - dir[c].name_offset) return lm_set_error(LM_ERR_IO, "bank class name outside the file");
    return LM_OK;
}

}  // namespace

extern "C" int lm_bank_file_info(const char* path, int32_t* pyramid_levels, int32_t* num_classes, int64_t* num_pyramids, int64_t* num_features) {
    if (!path) return lm_set_error(LM_ERR_INVALID, "null argument");
    Mapping m; BankHeader h; const ClassRec* dir = nullptr;
    int rc = map_bank(path, m, h, dir);
    if (rc) return rc;
    int64_t np = 0, nf = 0;
    for (uint32_t c = 0; c < h.num_classes; ++c) { np += dir[c].num_pyramids; nf += (int64_t)dir[c].num_features; }
    if (pyramid_levels) *pyramid_levels = (int32_t)h.pyramid_levels;
    if (num_classes) *num_classes = (int32_t)h.num_classes;
    if (num_pyramids) *num_pyramids = np;
    if (num_features) *num_features = nf;
    return LM_OK;
}

extern "C" int lm_bank_file_class_id(const char* path, int index, char* out, int capacity) {
    if (!path || !out || capacity <= 0) return lm_set_error(LM_ERR_INVALID, "bad argument");
    Mapping m; BankHeader h; const ClassRec* dir = nullptr;
    int rc = map_bank(path, m, h, dir);
    if (rc) return rc;
    if (index < 0 || (uint32_t)index >= h.num_classes) return lm_set_error(LM_ERR_INVALID, "class index out of range");
    if (dir[index].name_len >= (uint32_t)INT32_MAX || (int)dir[index].name_len >= capacity) return lm_set_error(LM_ERR_INVALID, "class id needs %u bytes", dir[index].name_len + 1);
    memcpy(out, m.p + dir[index].name_offset, dir[index].name_len);
    out[dir[index].name_len] = 0;
    return LM_OK;
}

// The classes named in class_ids (all when num_class_ids == 0) are added to the detector; like readClasses
// (LL.cpp:2059) a class that is already present is an error, and nothing is added unless every class validates.
extern "C" int lm_detector_read_bank(lm_detector* d, const char* path, const char* const* class_ids, int num_class_ids) {
    if (!d || !path || num_class_ids < 0 || (num_class_ids && !class_ids)) return lm_set_error(LM_ERR_INVALID, "bad argument");
    if (d->n_submitted != d->n_collected) return lm_set_error(LM_ERR_INVALID, "a frame is in flight: collect it first");
    Mapping m; BankHeader h; const ClassRec* dir = nullptr;
    int rc = map_bank(path, m, h, dir);
    if (rc) return rc;
    if ((int)h.pyramid_levels != d->pyramid_levels)
        return lm_set_error(LM_ERR_INVALID, "bank has %u pyramid levels, detector %d [LL.cpp:2052]", h.pyramid_levels, d->pyramid_levels);
    const size_t E = (size_t)d->pyramid_levels * 2;
    std::vector<uint32_t> pick;
    auto name_of = [&](uint32_t c) { return std::string((const char*)m.p + dir[c].name_offset, dir[c].name_len); };
    if (num_class_ids == 0) {
        for (uint32_t c = 0; c < h.num_classes; ++c) pick.push_back(c);
    } else {
        for (int i = 0; i < num_class_ids; ++i) {
            const std::string want = class_ids[i] ? class_ids[i] : "";
            uint32_t c = 0;
            while (c < h.num_classes && name_of(c) != want) ++c;
            if (c == h.num_classes) return lm_set_error(LM_ERR_NOT_FOUND, "class '%s' is not in %s", want.c_str(), path);
            pick.push_back(c);
        }
    }
    std::vector<std::pair<std::string, std::vector<TemplatePyramid>>> loaded;
    for (uint32_t c : pick) {
        const ClassRec& cr = dir[c];
        std::string cid = name_of(c);
        if (d->class_templates.count(cid)) return lm_set_error(LM_ERR_INVALID, "class '%s' already present [LL.cpp:2059]", cid.c_str());
        for (auto& kv : loaded) if (kv.first == cid) return lm_set_error(LM_ERR_INVALID, "class '%s' named twice", cid.c_str());
        const uint64_t nrec = (uint64_t)cr.num_pyramids * E + 1;
        // file-supplied counts: bounded by the file size BEFORE they are multiplied (a crafted num_features >= 2^62 would wrap)
        if (cr.num_features > m.n / sizeof(uint32_t) || nrec > m.n / sizeof(TemplRec) ||
            !m.span(cr.templ_offset, nrec * sizeof(TemplRec)) || !m.span(cr.feat_offset, cr.num_features * sizeof(uint32_t)))
            return lm_set_error(LM_ERR_IO, "arrays of class '%s' outside the file", cid.c_str());
        const TemplRec* recs = (const TemplRec*)(m.p + cr.templ_offset);
        const uint32_t* feats = (const uint32_t*)(m.p + cr.feat_offset);
        if (recs[0].feat_begin != 0 || recs[nrec - 1].feat_begin != cr.num_features)
            return lm_set_error(LM_ERR_IO, "feature index of class '%s' inconsistent", cid.c_str());
        std::vector<TemplatePyramid> tps((size_t)cr.num_pyramids);
        for (size_t p = 0; p < tps.size(); ++p) {
            TemplatePyramid& tp = tps[p];
            tp.resize(E);
            for (size_t e = 0; e < E; ++e) {
                const TemplRec& r = recs[p * E + e];
                const uint64_t a = r.feat_begin, b = recs[p * E + e + 1].feat_begin;
                if (b < a || b > cr.num_features) return lm_set_error(LM_ERR_IO, "feature index of class '%s' not monotone", cid.c_str());
                if (b - a > 8191) return lm_set_error(LM_ERR_INVALID, "templ.features.size() <= 8191 [LL.cpp:1291]");
                Template& t = tp[e];
                t.width = r.width; t.height = r.height; t.pyramid_level = (int)(e / 2);
                t.features.resize((size_t)(b - a));
                for (uint64_t i = a; i < b; ++i) t.features[(size_t)(i - a)] = unpack_feature(feats[i]);
            }
        }
        loaded.emplace_back(std::move(cid), std::move(tps));
    }
    for (auto& kv : loaded) d->class_templates[kv.first] = std::move(kv.second);
    if (!loaded.empty()) d->bank_dirty = true;
    return LM_OK;
}
