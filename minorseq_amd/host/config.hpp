// config.hpp — target configuration (doc/JULIET.md:109-180, 306-340): genes, DRM position grammar,
// optional referenceSequence, --region subsetting (:270-271); codon translation.
#pragma once
#include <algorithm>
#include <cctype>
#include <fstream>
#include <sstream>

#include "../../include/juliet_hip.h"
#include "json.hpp"

namespace jlhost {

inline char translate(unsigned codon)  // 16*b0 + 4*b1 + b2 over ACGT; stop = 'X' (R8X)
{
    static const char *tbl = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVXYXYSSSSXCWCLFLF";
    return codon < 64 ? tbl[codon] : '?';
}
inline std::string codon_string(unsigned c)
{
    static const char *b = "ACGT";
    return std::string{b[(c >> 4) & 3], b[(c >> 2) & 3], b[c & 3]};
}
inline uint8_t base_code(char ch)
{
    switch (std::toupper((unsigned char)ch)) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': case 'U': return 3;
    default: return 4;
    }
}

// one entry of drms/positions: "103", "M130", "M103L", "M103LKA", "103L", "103LG" (doc/JULIET.md:167-176)
struct DrmPosition {
    char ref_aa = '*';
    uint32_t pos = 0;
    std::string muts;  // empty = wildcard
    static DrmPosition parse(const std::string &s)
    {
        DrmPosition d;
        size_t i = 0;
        if (i < s.size() && std::isalpha((unsigned char)s[i])) d.ref_aa = (char)std::toupper((unsigned char)s[i++]);
        const size_t d0 = i;
        while (i < s.size() && std::isdigit((unsigned char)s[i])) ++i;
        if (i == d0) throw std::runtime_error("DRM position without a number: '" + s + "'");
        d.pos = (uint32_t)std::stoul(s.substr(d0, i - d0));
        for (; i < s.size(); ++i) {
            if (!std::isalpha((unsigned char)s[i]) && s[i] != '*') throw std::runtime_error("bad DRM position: '" + s + "'");
            d.muts += (char)std::toupper((unsigned char)s[i]);
        }
        return d;
    }
    bool matches(uint32_t aa_pos, char aa) const
    {
        return aa_pos == pos && (muts.empty() || muts.find('*') != std::string::npos || muts.find(aa) != std::string::npos);
    }
};

struct Drm {
    std::string name;
    std::vector<std::string> raw;
    std::vector<DrmPosition> positions;
};

struct GeneCfg {
    std::string name;
    uint32_t begin = 0, end = 0;  // 1-based [begin, end), as configured
    std::vector<Drm> drms;
    // after --region: the part actually called, and the index of its first codon within the gene
    uint32_t begin_eff = 0, end_eff = 0, first_codon = 0;
};

struct TargetConfig {
    std::vector<GeneCfg> genes;
    std::string reference_name, reference_sequence, version, database_version;
    bool from_user = false;

    static TargetConfig from_json(const Json &j)
    {
        TargetConfig c;
        c.from_user = true;
        c.reference_name = j.get_str("referenceName");
        c.reference_sequence = j.get_str("referenceSequence");
        c.version = j.get_str("version");
        c.database_version = j.get_str("databaseVersion");
        const Json *genes = j.get("genes");
        if (!genes || genes->type != Json::Array) throw std::runtime_error("target config: 'genes' array missing");
        for (const Json &g : genes->arr) {
            GeneCfg gc;
            gc.name = g.get_str("name", "unknown");
            const Json *b = g.get("begin"), *e = g.get("end");
            if (!b || !e || b->type != Json::Number || e->type != Json::Number)
                throw std::runtime_error("target config: gene '" + gc.name + "' needs numeric begin and end");
            gc.begin = (uint32_t)b->num;
            gc.end = (uint32_t)e->num;
            if (gc.begin < 1 || gc.end <= gc.begin) throw std::runtime_error("target config: gene '" + gc.name + "' has an empty range");
            gc.begin_eff = gc.begin;
            gc.end_eff = gc.end;
            if (const Json *drms = g.get("drms")) {
                for (const Json &d : drms->arr) {
                    Drm dr;
                    dr.name = d.get_str("name");
                    if (const Json *ps = d.get("positions"))
                        for (const Json &p : ps->arr) {
                            dr.raw.push_back(p.str);
                            dr.positions.push_back(DrmPosition::parse(p.str));
                        }
                    gc.drms.push_back(std::move(dr));
                }
            }
            c.genes.push_back(std::move(gc));
        }
        return c;
    }

    // `--config X`: a file, or a predefined name (doc/JULIET.md:118-126)
    static TargetConfig load(const std::string &arg)
    {
        if (arg == "ABL1") return from_json(Json::parse(abl1_json()));
        if (arg == "HIV" || arg == "HIV-PB")
            throw std::runtime_error("predefined config '" + arg + "' is not bundled: the reference documents only a 10-base "
                                     "prefix of its HXB2 sequence (doc/JULIET.md:154); pass a JSON file with --config");
        std::ifstream f(arg);
        if (!f) throw std::runtime_error("cannot read target config " + arg);
        std::stringstream ss;
        ss << f.rdbuf();
        return from_json(Json::parse(ss.str()));
    }

    // the BCR-ABL example of doc/JULIET.md:306-340 (no referenceSequence there => majority-codon mode)
    static std::string abl1_json()
    {
        return R"JSON({"genes":[{"name":"ABL1","begin":193,"end":3585,"drms":[
 {"name":"imatinib","positions":["T315AI","Y253H","E255KV","V299L","F317AICLV","F359CIV"]},
 {"name":"dasatinib","positions":["T315AI","V299L","F317AICLV"]},
 {"name":"nilotinib","positions":["T315AI","Y253H","E255KV","F359CIV"]},
 {"name":"bosutinib","positions":["T315AI"]}]}],
 "referenceName":"NM_005157.5","version":"Predefined ABL1 (doc/JULIET.md:306-340)"})JSON";
    }

    // --region b-e, 1-based [b,e): intersect every gene, snapping inward to the gene's own frame
    void apply_region(uint32_t rb, uint32_t re)
    {
        std::vector<GeneCfg> out;
        for (GeneCfg g : genes) {
            uint32_t b = std::max(g.begin, rb), e = std::min(g.end, re);
            if (b >= e) continue;
            const uint32_t off = (b - g.begin) % 3;
            if (off) b += 3 - off;
            if (b >= e) continue;
            g.first_codon = (b - g.begin) / 3;
            g.begin_eff = b;
            g.end_eff = e;
            out.push_back(std::move(g));
        }
        genes.swap(out);
        region_applied = true;
    }
    bool region_applied = false;

    std::string known_drms(size_t gene, uint32_t aa_pos, char aa) const
    {
        std::string s;
        for (const Drm &d : genes[gene].drms)
            for (const DrmPosition &p : d.positions)
                if (p.matches(aa_pos, aa)) {
                    if (!s.empty()) s += " + ";
                    s += d.name;
                    break;
                }
        return s;
    }

    Json echo() const  // "Target Config" section (doc/JULIET.md:83-88)
    {
        Json j = Json::object();
        j.set("version", Json::of(version));
        j.set("referenceName", Json::of(reference_name));
        j.set("referenceLength", Json::of((uint32_t)reference_sequence.size()));
        j.set("databaseVersion", Json::of(database_version));
        Json gs = Json::array();
        for (const GeneCfg &g : genes) {
            Json gj = Json::object();
            gj.set("name", Json::of(g.name)).set("begin", Json::of(g.begin)).set("end", Json::of(g.end));
            Json ds = Json::array();
            for (const Drm &d : g.drms) {
                Json dj = Json::object();
                dj.set("name", Json::of(d.name));
                Json ps = Json::array();
                for (const std::string &r : d.raw) ps.push(Json::of(r));
                dj.set("positions", ps);
                ds.push(dj);
            }
            gj.set("drms", ds);
            gs.push(gj);
        }
        j.set("genes", gs);
        return j;
    }
};

}  // namespace jlhost
