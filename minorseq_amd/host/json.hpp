// json.hpp — small JSON DOM (parse + write) for target configs in and result files out
// (doc/JULIET.md:129-157 config schema; :61-69 JSON output).
#pragma once
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace jlhost {

struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;  // insertion order kept for stable output

    static Json object() { Json j; j.type = Object; return j; }
    static Json array() { Json j; j.type = Array; return j; }
    static Json of(const std::string &s) { Json j; j.type = String; j.str = s; return j; }
    static Json of(const char *s) { return of(std::string(s)); }
    static Json of(double d) { Json j; j.type = Number; j.num = d; return j; }
    static Json of(int64_t d) { Json j; j.type = Number; j.num = (double)d; return j; }
    static Json of(uint32_t d) { Json j; j.type = Number; j.num = (double)d; return j; }
    static Json of(bool v) { Json j; j.type = Bool; j.b = v; return j; }

    Json &set(const std::string &k, Json v)
    {
        for (auto &kv : obj)
            if (kv.first == k) { kv.second = std::move(v); return *this; }
        obj.emplace_back(k, std::move(v));
        return *this;
    }
    Json &push(Json v) { arr.push_back(std::move(v)); return *this; }
    const Json *get(const std::string &k) const
    {
        for (auto &kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    std::string get_str(const std::string &k, const std::string &d = "") const
    {
        const Json *j = get(k);
        return j && j->type == String ? j->str : d;
    }

    void write(std::string &out, int indent = 0, int step = 1) const
    {
        const std::string pad((size_t)(indent + step), ' '), pad0((size_t)indent, ' ');
        switch (type) {
        case Null: out += "null"; break;
        case Bool: out += b ? "true" : "false"; break;
        case Number: {
            char buf[40];
            if (std::isfinite(num) && num == std::floor(num) && std::fabs(num) < 9e15) snprintf(buf, sizeof buf, "%.0f", num);
            else if (std::isfinite(num)) snprintf(buf, sizeof buf, "%.17g", num);
            else snprintf(buf, sizeof buf, "null");  // -inf log-p etc.
            out += buf;
            break;
        }
        case String: escape(out, str); break;
        case Array:
            if (arr.empty()) { out += "[]"; break; }
            out += "[\n";
            for (size_t i = 0; i < arr.size(); ++i) {
                out += pad;
                arr[i].write(out, indent + step, step);
                out += i + 1 < arr.size() ? ",\n" : "\n";
            }
            out += pad0 + "]";
            break;
        case Object:
            if (obj.empty()) { out += "{}"; break; }
            out += "{\n";
            for (size_t i = 0; i < obj.size(); ++i) {
                out += pad;
                escape(out, obj[i].first);
                out += ": ";
                obj[i].second.write(out, indent + step, step);
                out += i + 1 < obj.size() ? ",\n" : "\n";
            }
            out += pad0 + "}";
            break;
        }
    }

    static void escape(std::string &out, const std::string &s)
    {
        out += '"';
        for (unsigned char c : s) {
            switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\t': out += "\\t"; break;
            case '\r': out += "\\r"; break;
            default:
                if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
                else out += (char)c;
            }
        }
        out += '"';
    }

    static Json parse(const std::string &text)
    {
        size_t i = 0;
        Json j = parse_value(text, i);
        skip(text, i);
        if (i != text.size()) throw std::runtime_error("JSON: trailing characters");
        return j;
    }

private:
    static void skip(const std::string &t, size_t &i)
    {
        while (i < t.size() && (t[i] == ' ' || t[i] == '\n' || t[i] == '\t' || t[i] == '\r')) ++i;
    }
    static Json parse_value(const std::string &t, size_t &i)
    {
        skip(t, i);
        if (i >= t.size()) throw std::runtime_error("JSON: unexpected end");
        const char c = t[i];
        if (c == '{') {
            Json j = object();
            ++i;
            skip(t, i);
            if (i < t.size() && t[i] == '}') { ++i; return j; }
            for (;;) {
                skip(t, i);
                Json k = parse_string(t, i);
                skip(t, i);
                if (i >= t.size() || t[i] != ':') throw std::runtime_error("JSON: ':' expected");
                ++i;
                j.obj.emplace_back(k.str, parse_value(t, i));
                skip(t, i);
                if (i < t.size() && t[i] == ',') { ++i; continue; }
                if (i < t.size() && t[i] == '}') { ++i; return j; }
                throw std::runtime_error("JSON: ',' or '}' expected");
            }
        }
        if (c == '[') {
            Json j = array();
            ++i;
            skip(t, i);
            if (i < t.size() && t[i] == ']') { ++i; return j; }
            for (;;) {
                j.arr.push_back(parse_value(t, i));
                skip(t, i);
                if (i < t.size() && t[i] == ',') { ++i; continue; }
                if (i < t.size() && t[i] == ']') { ++i; return j; }
                throw std::runtime_error("JSON: ',' or ']' expected");
            }
        }
        if (c == '"') return parse_string(t, i);
        if (t.compare(i, 4, "true") == 0) { i += 4; return of(true); }
        if (t.compare(i, 5, "false") == 0) { i += 5; return of(false); }
        if (t.compare(i, 4, "null") == 0) { i += 4; return Json(); }
        size_t n = 0;
        double v;
        try { v = std::stod(t.substr(i, 64), &n); } catch (...) { throw std::runtime_error("JSON: bad value"); }
        i += n;
        return of(v);
    }
    static Json parse_string(const std::string &t, size_t &i)
    {
        if (i >= t.size() || t[i] != '"') throw std::runtime_error("JSON: string expected");
        ++i;
        std::string s;
        while (i < t.size() && t[i] != '"') {
            if (t[i] == '\\' && i + 1 < t.size()) {
                const char e = t[i + 1];
                i += 2;
                switch (e) {
                case 'n': s += '\n'; break;
                case 't': s += '\t'; break;
                case 'r': s += '\r'; break;
                case 'b': s += '\b'; break;
                case 'f': s += '\f'; break;
                case 'u': {
                    unsigned cp = (unsigned)std::stoul(t.substr(i, 4), nullptr, 16);
                    i += 4;
                    if (cp < 0x80) s += (char)cp;
                    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
                    else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: s += e;
                }
            } else {
                s += t[i++];
            }
        }
        if (i >= t.size()) throw std::runtime_error("JSON: unterminated string");
        ++i;
        return of(s);
    }
};

}  // namespace jlhost
