#include "mini_json.h"
#include <cstring>
#include <cstdio>
#include <cmath>
#include <climits>

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
		bool integral = true;
		for (const char* c = start; c < end; ++c) if (*c == '.' || *c == 'e' || *c == 'E') integral = false;
		p += (size_t)(end - start);
		if (integral && std::fabs(v) < 9.0e15) return Json((long long)v);
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
		case Binary: out = "\"<binary " + std::to_string(m_bin->size()) + " bytes>\""; break;
	}
	return out;
}

// ---- MessagePack -------------------------------------------------------------------------------------------------------------
// Writer choices follow nlohmann::json::to_msgpack (what the reference calls): smallest integer format, float32 when the value
// survives the round trip through float, str8/16/32, bin8/16/32 without ext subtype, maps in key order (std::map, like nlohmann).
namespace {
void put_be(std::string& o, uint64_t v, int n) { for (int i = n - 1; i >= 0; --i) o += (char)((v >> (8 * i)) & 0xff); }
void write_mp(const Json& j, std::string& o);
void write_str(const std::string& s, std::string& o) {
	const size_t n = s.size();
	if (n <= 31) o += (char)(0xa0 | n);
	else if (n <= 0xff) { o += (char)0xd9; put_be(o, n, 1); }
	else if (n <= 0xffff) { o += (char)0xda; put_be(o, n, 2); }
	else { o += (char)0xdb; put_be(o, n, 4); }
	o += s;
}
void write_mp(const Json& j, std::string& o) {
	switch (j.type()) {
		case Json::Null: o += (char)0xc0; break;
		case Json::Bool: o += (char)(j.boolean() ? 0xc3 : 0xc2); break;
		case Json::Number: {
			const double d = j.number();
			if (j.is_integer()) {
				if (d >= 0) {
					const uint64_t u = (uint64_t)d;
					if (u < 128) o += (char)u;
					else if (u <= 0xff) { o += (char)0xcc; put_be(o, u, 1); }
					else if (u <= 0xffff) { o += (char)0xcd; put_be(o, u, 2); }
					else if (u <= 0xffffffffull) { o += (char)0xce; put_be(o, u, 4); }
					else { o += (char)0xcf; put_be(o, u, 8); }
				} else {
					const int64_t i = (int64_t)d;
					if (i >= -32) o += (char)(int8_t)i;
					else if (i >= INT8_MIN) { o += (char)0xd0; put_be(o, (uint64_t)i, 1); }
					else if (i >= INT16_MIN) { o += (char)0xd1; put_be(o, (uint64_t)i, 2); }
					else if (i >= INT32_MIN) { o += (char)0xd2; put_be(o, (uint64_t)i, 4); }
					else { o += (char)0xd3; put_be(o, (uint64_t)i, 8); }
				}
			} else {
				const float f = (float)d;
				if ((double)f == d) { uint32_t u; memcpy(&u, &f, 4); o += (char)0xca; put_be(o, u, 4); }
				else { uint64_t u; memcpy(&u, &d, 8); o += (char)0xcb; put_be(o, u, 8); }
			}
			break;
		}
		case Json::String: write_str(j.str(), o); break;
		case Json::Binary: {
			const auto& b = j.bin();
			const size_t n = b.size();
			if (n <= 0xff) { o += (char)0xc4; put_be(o, n, 1); }
			else if (n <= 0xffff) { o += (char)0xc5; put_be(o, n, 2); }
			else if (n <= 0xffffffffull) { o += (char)0xc6; put_be(o, n, 4); }
			else throw std::runtime_error("msgpack: binary value larger than 4 GiB");
			o.append((const char*)b.data(), n);
			break;
		}
		case Json::Array: {
			const size_t n = j.size();
			if (n <= 15) o += (char)(0x90 | n);
			else if (n <= 0xffff) { o += (char)0xdc; put_be(o, n, 2); }
			else { o += (char)0xdd; put_be(o, n, 4); }
			for (const Json& e : j.elements()) write_mp(e, o);
			break;
		}
		case Json::Object: {
			const size_t n = j.size();
			if (n <= 15) o += (char)(0x80 | n);
			else if (n <= 0xffff) { o += (char)0xde; put_be(o, n, 2); }
			else { o += (char)0xdf; put_be(o, n, 4); }
			for (const auto& kv : j.items()) { write_str(kv.first, o); write_mp(kv.second, o); }
			break;
		}
	}
}

struct MpReader {
	const uint8_t* d; size_t n, p = 0;
	void need(size_t k) const { if (p + k > n) throw std::runtime_error("msgpack: unexpected end of input"); }
	uint64_t be(int k) { need(k); uint64_t v = 0; for (int i = 0; i < k; ++i) v = (v << 8) | d[p++]; return v; }
	std::string str(size_t k) { need(k); std::string s((const char*)d + p, k); p += k; return s; }
	Json bin(size_t k) { need(k); Json j = Json::binary(d + p, k); p += k; return j; }
	Json arr(size_t k) { Json j = Json::array(); for (size_t i = 0; i < k; ++i) j.push_back(value()); return j; }
	Json map(size_t k) {
		Json j = Json::object();
		for (size_t i = 0; i < k; ++i) {
			Json key = value();
			if (!key.is_string()) throw std::runtime_error("msgpack: map key is not a string");
			j[key.str()] = value();
		}
		return j;
	}
	Json value() {
		need(1);
		const uint8_t c = d[p++];
		if (c <= 0x7f) return Json((long long)c);
		if (c >= 0xe0) return Json((long long)(int8_t)c);
		if ((c & 0xf0) == 0x80) return map(c & 0x0f);
		if ((c & 0xf0) == 0x90) return arr(c & 0x0f);
		if ((c & 0xe0) == 0xa0) return Json(str(c & 0x1f));
		switch (c) {
			case 0xc0: return Json();
			case 0xc2: return Json(false);
			case 0xc3: return Json(true);
			case 0xc4: return bin(be(1));
			case 0xc5: return bin(be(2));
			case 0xc6: return bin(be(4));
			case 0xca: { uint32_t u = (uint32_t)be(4); float f; memcpy(&f, &u, 4); return Json((double)f); }
			case 0xcb: { uint64_t u = be(8); double f; memcpy(&f, &u, 8); return Json(f); }
			case 0xcc: return Json((long long)be(1));
			case 0xcd: return Json((long long)be(2));
			case 0xce: return Json((long long)be(4));
			case 0xcf: return Json((unsigned long long)be(8));
			case 0xd0: return Json((long long)(int8_t)be(1));
			case 0xd1: return Json((long long)(int16_t)be(2));
			case 0xd2: return Json((long long)(int32_t)be(4));
			case 0xd3: return Json((long long)(int64_t)be(8));
			case 0xd9: return Json(str(be(1)));
			case 0xda: return Json(str(be(2)));
			case 0xdb: return Json(str(be(4)));
			case 0xdc: return arr(be(2));
			case 0xdd: return arr(be(4));
			case 0xde: return map(be(2));
			case 0xdf: return map(be(4));
			default: throw std::runtime_error("msgpack: unsupported type byte (ext / reserved)");
		}
	}
};
} // namespace

std::string Json::to_msgpack() const { std::string o; write_mp(*this, o); return o; }

Json Json::from_msgpack(const void* data, size_t n_bytes) {
	MpReader r{(const uint8_t*)data, n_bytes};
	Json j = r.value();
	if (r.p != n_bytes) throw std::runtime_error("msgpack: trailing bytes after the top-level value");
	return j;
}

Json Json::from_msgpack_file(const std::string& path) {
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) throw std::runtime_error("cannot open " + path);
	std::string buf;
	char tmp[1 << 16];
	size_t k;
	while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, k);
	fclose(f);
	return from_msgpack(buf.data(), buf.size());
}

void Json::to_msgpack_file(const std::string& path) const {
	const std::string o = to_msgpack();
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) throw std::runtime_error("cannot write " + path);
	const size_t k = fwrite(o.data(), 1, o.size(), f);
	fclose(f);
	if (k != o.size()) throw std::runtime_error("short write to " + path);
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
