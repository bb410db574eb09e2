// html.hpp — the HTML output: "a 1:1 conversion of the JSON file [that] contains the identical information, but more
// human-readable" (doc/JULIET.md:68-69), in the four sections of doc/JULIET.md:71-107 (Input data, Target config,
// Variant Discovery with the -3..+5 context counts and, with --mode-phasing, the haplotype columns, Drug Summaries)
// plus the root `haplotype` block (read categories of the tooltips, doc/JULIET.md:372-389; counts and read names,
// :209-211).  Every leaf of the JSON object is rendered; tests/test_gpu_cli.py parses the page back and compares it
// cell by cell.  Numbers are displayed as the reference's screenshots print them (format.hpp).
#pragma once
#include <string>

#include "format.hpp"
#include "json.hpp"

namespace jlhost {

inline std::string html_escape(const std::string &s)
{
    std::string o;
    for (char c : s) {
        if (c == '<') o += "&lt;";
        else if (c == '>') o += "&gt;";
        else if (c == '&') o += "&amp;";
        else if (c == '"') o += "&quot;";
        else o += c;
    }
    return o;
}

inline std::string num_str(const Json *j)
{
    if (!j) return "";
    if (j->type == Json::String) return j->str;
    if (j->type == Json::Bool) return j->b ? "true" : "false";
    std::string s;
    j->write(s);
    return s;
}

inline std::string td(const std::string &s, const char *cls = nullptr)
{
    return std::string("<td") + (cls ? std::string(" class=\"") + cls + "\"" : std::string()) + ">" + html_escape(s) + "</td>";
}

inline std::string render_html(const Json &root)
{
    std::string h = "<!DOCTYPE html><html><head><meta charset=\"utf-8\"><title>Minor Variants Summary (Juliet)</title>"
                    "<style>body{font-family:sans-serif}table{border-collapse:collapse;margin:4px 0}td,th{border:1px solid #999;padding:2px 8px;"
                    "text-align:center}td.hit{background:#e8381b;color:#fff}td.nohit{background:#4a4a4a}summary{font-weight:bold;font-size:120%}"
                    "table.msa td.wt{font-weight:bold}</style></head><body>\n<h1>Minor Variants Summary (Juliet)</h1>\n";
    // ---- 1. Input data (doc/JULIET.md:73-79)
    if (const Json *in = root.get("input")) {
        h += "<details open id=\"input\"><summary>Input data</summary><table id=\"input-table\">\n";
        for (auto &kv : in->obj) h += "<tr><th>" + html_escape(kv.first) + "</th>" + td(num_str(&kv.second)) + "</tr>\n";
        h += "</table></details>\n";
    }
    // ---- 2. Target config (doc/JULIET.md:83-88)
    if (const Json *tc = root.get("target_config")) {
        h += "<details open id=\"target\"><summary>Target config</summary><table id=\"target-table\">\n";
        for (auto &kv : tc->obj)
            if (kv.first != "genes") h += "<tr><th>" + html_escape(kv.first) + "</th>" + td(num_str(&kv.second)) + "</tr>\n";
        h += "</table>\n<ul id=\"target-genes\">\n";
        if (const Json *gs = tc->get("genes"))
            for (const Json &g : gs->arr) {
                h += "<li data-begin=\"" + num_str(g.get("begin")) + "\" data-end=\"" + num_str(g.get("end")) + "\"><b>" +
                     html_escape(g.get_str("name")) + "</b> (" + num_str(g.get("begin")) + "-" + num_str(g.get("end")) + ")";
                const Json *ds = g.get("drms");
                if (ds && !ds->arr.empty()) {
                    h += "<ul>";
                    for (const Json &d : ds->arr) {
                        h += "<li class=\"drm\"><span class=\"drm-name\">" + html_escape(d.get_str("name")) + "</span>:";
                        if (const Json *ps = d.get("positions"))
                            for (const Json &p : ps->arr) h += " <span class=\"drm-pos\">" + html_escape(p.str) + "</span>";
                        h += "</li>";
                    }
                    h += "</ul>";
                }
                h += "</li>\n";
            }
        h += "</ul></details>\n";
    }
    // haplotype header shared by every gene table: names and percentages (columns are global across genes,
    // juliet_hiv-phasing.png)
    const Json *hb = root.get("haplotype");
    const Json *haps = hb ? hb->get("haplotypes") : nullptr;
    // ---- 3. Variant Discovery (doc/JULIET.md:92-102)
    if (const Json *genes = root.get("genes")) {
        h += "<details open id=\"variants\"><summary>Variant Discovery</summary>\n";
        const std::string ref_name = root.get("target_config") ? root.get("target_config")->get_str("referenceName") : "";
        for (const Json &g : genes->arr) {
            h += "<table class=\"gene\" data-gene=\"" + html_escape(g.get_str("name")) + "\"><caption>" + html_escape(g.get_str("name")) +
                 "</caption>\n<tr><th colspan=\"3\">" + html_escape(ref_name.empty() ? "Majority Call" : ref_name) +
                 "</th><th colspan=\"5\">Sample Variants</th>";
            if (haps)
                for (const Json &hp : haps->arr) h += "<th class=\"hapname\">" + html_escape(hp.get_str("name")) + "</th>";
            h += "</tr>\n<tr><th>Codon</th><th>AA</th><th>Pos</th><th>AA</th><th>Codon</th><th>%</th><th>Coverage</th><th>Affected Drugs</th>";
            if (haps)
                for (const Json &hp : haps->arr) h += "<th class=\"happerc\">" + format_hap_percent(100.0 * hp.get("frequency")->num) + "</th>";
            h += "</tr>\n";
            if (const Json *vps = g.get("variant_positions"))
                for (const Json &vp : vps->arr) {
                    for (const Json &aa : vp.get("variant_amino_acids")->arr)
                        for (const Json &vc : aa.get("variant_codons")->arr) {
                            h += "<tr class=\"variant\" title=\"count=" + num_str(vc.get("count")) + " expected=" + num_str(vc.get("expected")) +
                                 " pValue=" + num_str(vc.get("pValue")) + " log_pValue=" + num_str(vc.get("log_pValue")) + "\">" +
                                 td(vp.get_str("ref_codon")) + td(vp.get_str("ref_amino_acid")) + td(num_str(vp.get("ref_position"))) +
                                 td(aa.get_str("amino_acid")) + td(vc.get_str("codon")) + td(format_percent(100.0 * vc.get("frequency")->num)) +
                                 td(num_str(vp.get("coverage"))) + td(vc.get_str("known_drm"));
                            if (const Json *hh = vc.get("haplotype_hit"))
                                for (const Json &x : hh->arr) h += x.b ? "<td class=\"hit\">x</td>" : "<td class=\"nohit\"></td>";
                            h += "</tr>\n";
                        }
                    // "Clicking the row will show counts of the multiple-sequence alignment counts of the -3 to +3 context positions"
                    if (const Json *msa = vp.get("msa")) {
                        h += "<tr class=\"context\"><td colspan=\"8\"><details><summary>context</summary><table class=\"msa\" data-pos=\"" +
                             num_str(vp.get("ref_position")) + "\"><tr><th>Pos</th><th>Abs</th><th>A</th><th>C</th><th>G</th><th>T</th><th>-</th><th>N</th><th>wt</th></tr>\n";
                        for (const Json &m : msa->arr) {
                            const std::string wt = m.get_str("wt");
                            h += "<tr>" + td(num_str(m.get("rel_pos"))) + td(num_str(m.get("abs_pos")));
                            static const char *sym[6] = {"A", "C", "G", "T", "-", "N"};
                            for (int s = 0; s < 6; ++s) h += td(num_str(m.get(sym[s])), wt == sym[s] ? "wt" : nullptr);
                            h += td(wt) + "</tr>\n";
                        }
                        h += "</table></details></td></tr>\n";
                    }
                }
            h += "</table>\n";
        }
        h += "</details>\n";
    }
    // ---- 4. Drug Summaries (doc/JULIET.md:104-107)
    if (const Json *ds = root.get("drug_summaries")) {
        h += "<details open id=\"drugs\"><summary>Drug Summaries</summary><table id=\"drug-table\"><tr><th>Drug</th><th>Gene</th><th>Mutation</th><th>Codon</th><th>%</th></tr>\n";
        for (const Json &d : ds->arr)
            for (const Json &v : d.get("variants")->arr)
                h += "<tr>" + td(d.get_str("drug")) + td(v.get_str("gene")) + td(v.get_str("mutation")) + td(v.get_str("codon")) +
                     td(format_percent(100.0 * v.get("frequency")->num)) + "</tr>\n";
        h += "</table></details>\n";
    }
    // ---- the root `haplotype` block (doc/JULIET.md:209-211, 372-389)
    if (hb) {
        h += "<details open id=\"haplotypes\"><summary>Haplotypes</summary><table id=\"hap-categories\"><tr><th>Haplotype Category</th><th>#Reads</th></tr>\n";
        static const char *cat[6][2] = {{"reported_reads", "Reported"}, {"insufficient_coverage_reads", "Insufficient Coverage (unreported)"},
                                        {"damaged_reads", "Overall Damaged (unreported)"}, {"marginal_gaps", "- Marginal Gaps"},
                                        {"marginal_heteroduplexes", "- Marginal Heteroduplexes"}, {"marginal_partial", "- Marginal Partial"}};
        for (auto &c : cat) h += "<tr data-key=\"" + std::string(c[0]) + "\"><th>" + c[1] + "</th>" + td(num_str(hb->get(c[0]))) + "</tr>\n";
        h += "</table>\n<p id=\"hap-positions\">";
        if (const Json *pc = hb->get("variant_positions_abs"))
            for (const Json &p : pc->arr) h += "<span>" + num_str(&p) + "</span> ";
        h += "</p>\n<table id=\"hap-table\"><tr><th>Haplotype</th><th>%</th><th>#Reads</th><th>Codons</th><th>Read names</th></tr>\n";
        if (haps)
            for (const Json &hp : haps->arr) {
                h += "<tr>" + td(hp.get_str("name")) + td(format_hap_percent(100.0 * hp.get("frequency")->num)) + td(num_str(hp.get("reads"))) + "<td>";
                if (const Json *cs = hp.get("codons"))
                    for (size_t i = 0; i < cs->arr.size(); ++i) h += (i ? " " : "") + cs->arr[i].str;
                h += "</td><td><details><summary>" + std::to_string(hp.get("read_names") ? hp.get("read_names")->arr.size() : 0) + "</summary>";
                if (const Json *rn = hp.get("read_names"))
                    for (const Json &r : rn->arr) h += "<span class=\"rn\">" + html_escape(r.str) + "</span> ";
                h += "</details></td></tr>\n";
            }
        h += "</table></details>\n";
    }
    h += "</body></html>\n";
    return h;
}

}  // namespace jlhost
