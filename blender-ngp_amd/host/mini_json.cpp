#include "mini_json.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace ngp {

class JsonParser {
public:
	explicit JsonParser(const std::string& t) : s(t), p(0) {}
	Json parse_document() {
		Json v = parse_value();
		skip_ws();
		if (p != s.size()) fail("trailing characters");
		return v;
	}

private:
	const std::string& s;
	size_t p;

	[[noreturn]] void fail(const char* what) const {
		throw std::runtime_error(std::string("json parse error at offset ") + std::to_string(p) + ": " + what);
	}
	void skip_ws() {
		while (p < s.size()) {
			char c = s[p];
			if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { ++p; continue; }
			if (c == '/' && p + 1 < s.size() && s[p + 1] == '/') { while (p < s.size() && s[p] != '\n') ++p; continue; }
			if (c == '/' && p + 1 < s.size() && s[p + 1] == '*') {
				p += 2;
				while (p + 1 < s.size() && !(s[p] == '*' && s[p + 1] == '/')) ++p;
				p = p + 2 <= s.size() ? p + 2 : s.size();
				continue;
			}
			break;
		}
	}
	Json parse_value() {
		skip_ws();
		if (p >= s.size()) fail("unexpected end");
		char c = s[p];
		if (c == '{') return parse_object();
		if (c == '[') return parse_array();
		if (c == '"') return Json(parse_string());
		if (s.compare(p, 4, "true") == 0) { p += 4; return Json(true); }
		if (s.compare(p, 5, "false") == 0) { p += 5; return Json(false); }
		if (s.compare(p, 4, "null") == 0) { p += 4; return Json(); }
		return parse_number();
	}
	Json parse_number() {
		const char* start = s.c_str() + p;
		char* end = nullptr;
		double v = std::strtod(start, &end);
		if (end == start) fail("invalid value");
		p += (size_t)(end - start);
		return Json(v);
	}
	std::string parse_string() {
		std::string out;
		++p; // opening quote
		while (p < s.size() && s[p] != '"') {
			char c = s[p++];
			if (c == '\\') {
				if (p >= s.size()) fail("bad escape");
				char e = s[p++];
				switch (e) {
					case 'n': out += '\n'; break;
					case 't': out += '\t'; break;
					case 'r': out += '\r'; break;
					case 'b': out += '\b'; break;
					case 'f': out += '\f'; break;
					case 'u': {
						if (p + 4 > s.size()) fail("bad \\u escape");
						unsigned code = (unsigned)std::strtoul(s.substr(p, 4).c_str(), nullptr, 16);
						p += 4;
						if (code < 0x80) out += (char)code;
						else if (code < 0x800) { out += (char)(0xC0 | (code >> 6)); out += (char)(0x80 | (code & 0x3F)); }
						else { out += (char)(0xE0 | (code >> 12)); out += (char)(0x80 | ((code >> 6) & 0x3F)); out += (char)(0x80 | (code & 0x3F)); }
						break;
					}
					default: out += e; break;
				}
			} else {
				out += c;
			}
		}
		if (p >= s.size()) fail("unterminated string");
		++p;
		return out;
	}
	Json parse_array() {
		Json a = Json::array();
		++p;
		skip_ws();
		if (p < s.size() && s[p] == ']') { ++p; return a; }
		while (true) {
			a.m_arr.push_back(parse_value());
			skip_ws();
			if (p < s.size() && s[p] == ',') { ++p; skip_ws(); if (p < s.size() && s[p] == ']') { ++p; return a; } continue; }
			if (p < s.size() && s[p] == ']') { ++p; return a; }
			fail("expected , or ]");
		}
	}
	Json parse_object() {
		Json o = Json::object();
		++p;
		skip_ws();
		if (p < s.size() && s[p] == '}') { ++p; return o; }
		while (true) {
			skip_ws();
			if (p >= s.size() || s[p] != '"') fail("expected key");
			std::string k = parse_string();
			skip_ws();
			if (p >= s.size() || s[p] != ':') fail("expected :");
			++p;
			o.m_obj[k] = parse_value();
			skip_ws();
			if (p < s.size() && s[p] == ',') { ++p; skip_ws(); if (p < s.size() && s[p] == '}') { ++p; return o; } continue; }
			if (p < s.size() && s[p] == '}') { ++p; return o; }
			fail("expected , or }");
		}
	}
};

Json Json::parse(const std::string& text) { return JsonParser(text).parse_document(); }

Json Json::parse_file(const std::string& path) {
	std::ifstream f(path);
	if (!f) throw std::runtime_error("could not open " + path);
	std::stringstream ss;
	ss << f.rdbuf();
	return parse(ss.str());
}

static void dump_string(const std::string& s, std::string& out) {
	out += '"';
	for (char c : s) {
		switch (c) {
			case '"': out += "\\\""; break;
			case '\\': out += "\\\\"; break;
			case '\n': out += "\\n"; break;
			case '\t': out += "\\t"; break;
			case '\r': out += "\\r"; break;
			default: out += c; break;
		}
	}
	out += '"';
}

std::string Json::dump() const {
	std::string out;
	switch (m_type) {
		case Null: out = "null"; break;
		case Bool: out = m_bool ? "true" : "false"; break;
		case Number: {
			char buf[64];
			if (std::floor(m_num) == m_num && std::fabs(m_num) < 1e15) snprintf(buf, sizeof(buf), "%lld", (long long)m_num);
			else snprintf(buf, sizeof(buf), "%.17g", m_num);
			out = buf;
			break;
		}
		case String: dump_string(m_str, out); break;
		case Array: {
			out = "[";
			for (size_t i = 0; i < m_arr.size(); ++i) { if (i) out += ","; out += m_arr[i].dump(); }
			out += "]";
			break;
		}
		case Object: {
			out = "{";
			bool first = true;
			for (const auto& kv : m_obj) { if (!first) out += ","; first = false; dump_string(kv.first, out); out += ":"; out += kv.second.dump(); }
			out += "}";
			break;
		}
	}
	return out;
}

void Json::merge_patch(const Json& patch) {
	if (!patch.is_object()) { *this = patch; return; }
	if (!is_object()) *this = Json::object();
	for (const auto& kv : patch.m_obj) {
		if (kv.second.is_null()) m_obj.erase(kv.first);
		else m_obj[kv.first].merge_patch(kv.second);
	}
}

} // namespace ngp
