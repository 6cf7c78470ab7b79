// =============================================================================
// misp_io.cpp -- DIMACS-like .clq reader with the grammar and quirks of
// /root/reference/ddo/examples/misp/main.rs:258-317:
//   every line is trimmed, empty lines are skipped;
//   `^c\s.*$`                          comment
//   `^p\s+edge\s+(\d+)\s+(\d+)$`       n vertices: all rows = {0..n-1}, all weights = 1
//   `^n\s+(\d+)\s+(-?\d+)`             weight[id-1] = w          (prefix match)
//   `^e\s+(\d+)\s+(\d+)`               clear bit dst in row src and vice versa (1-based, prefix match)
//   anything else                      format error
// The rows are therefore COMPLEMENT adjacency rows and keep their own diagonal bit.
// =============================================================================
#include <cctype>
#include <cstdio>
#include <fstream>
#include <regex>
#include <sstream>
#include <string>
#include <vector>

#include "engine.hpp"

namespace ddo_hip {
namespace {
struct Cursor {
    const std::string& s;
    size_t i = 0;
    explicit Cursor(const std::string& s) : s(s) {}
    bool lit(char c) {
        if (i < s.size() && s[i] == c) { ++i; return true; }
        return false;
    }
    bool word(const char* w) {
        size_t k = i;
        for (; *w; ++w, ++k)
            if (k >= s.size() || s[k] != *w) return false;
        i = k;
        return true;
    }
    bool spaces1() {  // \s+
        size_t k = i;
        while (k < s.size() && std::isspace((unsigned char)s[k])) ++k;
        if (k == i) return false;
        i = k;
        return true;
    }
    bool digits(unsigned long long& out) {  // \d+
        size_t k = i;
        unsigned long long v = 0;
        while (k < s.size() && std::isdigit((unsigned char)s[k])) {
            v = v * 10 + (unsigned)(s[k] - '0');
            ++k;
        }
        if (k == i) return false;
        i = k;
        out = v;
        return true;
    }
    bool sdigits(long long& out) {  // -?\d+
        size_t save = i;
        bool neg = lit('-');
        unsigned long long v;
        if (!digits(v)) { i = save; return false; }
        out = neg ? -(long long)v : (long long)v;
        return true;
    }
    bool end() const { return i == s.size(); }
};
}  // namespace

bool read_misp_clq(const std::string& path, int& n, std::vector<uint64_t>& rows, std::vector<int64_t>& weights) {
    std::ifstream f(path);
    if (!f) {
        set_error("io error: cannot open " + path);
        return false;
    }
    n = 0;
    int ws = 0;
    rows.clear();
    weights.clear();
    std::string raw;
    size_t lineno = 0;
    while (std::getline(f, raw)) {
        ++lineno;
        size_t b = 0, e = raw.size();
        while (b < e && std::isspace((unsigned char)raw[b])) ++b;
        while (e > b && std::isspace((unsigned char)raw[e - 1])) --e;
        if (b == e) continue;
        const std::string line = raw.substr(b, e - b);
        {   // comment
            Cursor c(line);
            if (c.lit('c') && c.i < line.size() && std::isspace((unsigned char)line[c.i])) continue;
        }
        {   // problem declaration (anchored at both ends)
            Cursor c(line);
            unsigned long long nv, ne;
            if (c.lit('p') && c.spaces1() && c.word("edge") && c.spaces1() && c.digits(nv) && c.spaces1() && c.digits(ne) &&
                c.end()) {
                n = (int)nv;
                ws = (n + 63) / 64;
                rows.assign((size_t)n * ws, 0);
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) rows[(size_t)i * ws + j / 64] |= 1ULL << (j % 64);
                weights.assign(n, 1);
                continue;
            }
        }
        {   // node weight
            Cursor c(line);
            unsigned long long id;
            long long w;
            if (c.lit('n') && c.spaces1() && c.digits(id) && c.spaces1() && c.sdigits(w)) {
                if (id < 1 || id > (unsigned long long)n) {
                    set_error("ill formed instance: vertex id out of range at line " + std::to_string(lineno));
                    return false;
                }
                weights[id - 1] = w;
                continue;
            }
        }
        {   // edge
            Cursor c(line);
            unsigned long long a, d;
            if (c.lit('e') && c.spaces1() && c.digits(a) && c.spaces1() && c.digits(d)) {
                if (a < 1 || d < 1 || a > (unsigned long long)n || d > (unsigned long long)n) {
                    set_error("ill formed instance: vertex id out of range at line " + std::to_string(lineno));
                    return false;
                }
                size_t src = a - 1, dst = d - 1;
                rows[src * ws + dst / 64] &= ~(1ULL << (dst % 64));
                rows[dst * ws + src / 64] &= ~(1ULL << (src % 64));
                continue;
            }
        }
        set_error("ill formed instance (line " + std::to_string(lineno) + ")");
        return false;
    }
    if (n <= 0) {
        set_error("ill formed instance: no `p edge` line");
        return false;
    }
    return true;
}

bool read_knapsack(const std::string& path, int64_t& capacity, std::vector<int64_t>& profit, std::vector<int64_t>& weight) {
    std::ifstream f(path);
    if (!f) {
        set_error("cannot open " + path);
        return false;
    }
    profit.clear();
    weight.clear();
    std::string line;
    bool header = true;
    long long n = 0;
    while (std::getline(f, line)) {
        if (!line.empty() && line[0] == 'c') continue;   // main.rs:281-283
        std::istringstream is(line);
        long long a = 0, b = 0;
        if (header) {
            if (!(is >> a >> b)) continue;               // blank lines before the header
            n = a;
            capacity = b;
            header = false;
        } else {
            if ((long long)profit.size() >= n) break;   // main.rs:292: only the first n items count
            if (!(is >> a >> b)) continue;
            profit.push_back(a);
            weight.push_back(b);
        }
    }
    if (header || n < 1 || (long long)profit.size() != n) {
        set_error("malformed knapsack instance " + path);
        return false;
    }
    return true;
}

bool read_mcp(const std::string& path, int& n, std::vector<int64_t>& adj) {
    std::ifstream f(path);
    if (!f) {
        set_error("cannot open " + path);
        return false;
    }
    n = 0;
    adj.clear();
    std::string line;
    while (std::getline(f, line)) {
        size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        line = line.substr(b, e - b + 1);
        if (line.rfind("c ", 0) == 0) continue;                       // graph.rs:60-62
        // the two line shapes of graph.rs:49-50: all-digit tokens, nothing else on the line
        std::istringstream is(line);
        std::vector<std::string> tok;
        for (std::string t; is >> t;) tok.push_back(t);
        auto is_uint = [](const std::string& t) { return !t.empty() && t.find_first_not_of("0123456789") == std::string::npos; };
        auto is_int = [&](const std::string& t) { return !t.empty() && is_uint(t[0] == '-' ? t.substr(1) : t); };
        if (tok.size() == 2 && is_uint(tok[0]) && is_uint(tok[1])) {   // "<vertices> <edges>": a fresh graph
            n = (int)std::stol(tok[0]);
            adj.assign((size_t)n * (size_t)n, 0);
        } else if (tok.size() == 3 && is_uint(tok[0]) && is_uint(tok[1]) && is_int(tok[2])) {
            const long x = std::stol(tok[0]) - 1, y = std::stol(tok[1]) - 1;
            if (x < 0 || y < 0 || x >= n || y >= n) {
                set_error("edge outside the graph in " + path);
                return false;
            }
            adj[(size_t)x * n + y] = adj[(size_t)y * n + x] = std::stoll(tok[2]);   // add_bidir_edge
        }
    }
    if (n < 1) {
        set_error("malformed max-cut instance " + path);
        return false;
    }
    return true;
}

bool read_max2sat(const std::string& path, int& n, std::vector<int64_t>& lit_a, std::vector<int64_t>& lit_b, std::vector<int64_t>& weight) {
    std::ifstream f(path);
    if (!f) {
        set_error("cannot open " + path);
        return false;
    }
    // the four line shapes of data.rs:71-74, tried in the reference's order on the trimmed line (prefix matches)
    static const std::regex comment(R"(^c\s.*$)");
    static const std::regex pb_decl(R"(^p\s+wcnf\s+(\d+)\s+(\d+))");
    static const std::regex bin_decl(R"(^(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+0)");
    static const std::regex unit_decl(R"(^(-?\d+)\s+(-?\d+)-?\s+0)");
    n = 0;
    lit_a.clear();
    lit_b.clear();
    weight.clear();
    std::string line;
    while (std::getline(f, line)) {
        size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        line = line.substr(b, e - b + 1);
        std::smatch m;
        if (std::regex_search(line, m, comment)) continue;
        if (std::regex_search(line, m, pb_decl)) {
            n = (int)std::stol(m[1].str());
            continue;
        }
        if (std::regex_search(line, m, bin_decl)) {
            weight.push_back(std::stoll(m[1].str()));
            lit_a.push_back(std::stoll(m[2].str()));
            lit_b.push_back(std::stoll(m[3].str()));
            continue;
        }
        if (std::regex_search(line, m, unit_decl)) {
            weight.push_back(std::stoll(m[1].str()));
            lit_a.push_back(std::stoll(m[2].str()));
            lit_b.push_back(std::stoll(m[2].str()));
        }
    }
    if (n < 1) {
        set_error("malformed wcnf instance " + path);
        return false;
    }
    return true;
}

/// examples/tsptw/instance.rs:52-109.  Lines are trimmed; empty lines and lines starting with '#' are skipped; the first
/// remaining line starts with the number of nodes, the next nb_nodes lines are the rows of the distance matrix, every later
/// line is "earliest latest".  A number x becomes `(x as f32 * 10000.0) as usize`: single-precision product, truncated
/// towards zero, negative or NaN -> 0.
static int64_t tsptw_fixed(const std::string& tok) {
    const float v = std::strtof(tok.c_str(), nullptr);
    const float m = v * 10000.0f;
    if (!(m > 0.0f)) return 0;
    return (int64_t)m;
}
bool read_tsptw(const std::string& path, int& n, std::vector<int64_t>& dist, std::vector<int64_t>& earliest, std::vector<int64_t>& latest) {
    std::ifstream f(path);
    if (!f) {
        set_error("cannot open " + path);
        return false;
    }
    n = 0;
    dist.clear();
    earliest.clear();
    latest.clear();
    long lc = 0;
    std::string line;
    while (std::getline(f, line)) {
        const size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        line = line.substr(b, e - b + 1);
        if (line[0] == '#') continue;
        std::istringstream ss(line);
        std::string tok;
        if (lc == 0) {
            ss >> tok;
            n = (int)std::strtol(tok.c_str(), nullptr, 10);
            if (n < 1 || n > 65535) {
                set_error("malformed tsptw instance " + path);
                return false;
            }
            dist.assign((size_t)n * n, 0);
        } else if (lc <= n) {
            int j = 0;
            while (ss >> tok) {
                if (j < n) dist[(size_t)(lc - 1) * n + j] = tsptw_fixed(tok);
                ++j;
            }
        } else {
            std::string a, c;
            ss >> a >> c;
            earliest.push_back(tsptw_fixed(a));
            latest.push_back(tsptw_fixed(c));
        }
        lc += 1;
    }
    if (n < 1 || (int)earliest.size() < n) {
        set_error("malformed tsptw instance " + path);
        return false;
    }
    return true;
}

}  // namespace ddo_hip
