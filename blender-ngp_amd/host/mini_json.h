// mini_json — the small JSON subset the host needs (the reference takes nlohmann::json from inside tiny-cuda-nn's dependencies,
// which are absent).  Supports // and /* */ comments like load_network_config (src/testbed.cu:120-145),
// RFC 7386 merge_patch for the recursive "parent" inheritance (src/testbed.cu:77-88), binary values and the MessagePack
// encoding that `.msgpack` snapshots use (json::to_msgpack / from_msgpack, src/testbed.cu:3041, 139).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ngp {

class Json {
public:
	enum Type { Null, Bool, Number, String, Array, Object, Binary };
	Json() : m_type(Null) {}
	Json(bool b) : m_type(Bool), m_bool(b) {}
	Json(double d) : m_type(Number), m_num(d) {}
	Json(int d) : m_type(Number), m_num(d), m_is_int(true) {}
	Json(unsigned d) : m_type(Number), m_num(d), m_is_int(true) {}
	Json(long d) : m_type(Number), m_num((double)d), m_is_int(true) {}
	Json(unsigned long d) : m_type(Number), m_num((double)d), m_is_int(true) {}
	Json(long long d) : m_type(Number), m_num((double)d), m_is_int(true) {}
	Json(unsigned long long d) : m_type(Number), m_num((double)d), m_is_int(true) {}
	Json(const std::string& s) : m_type(String), m_str(s) {}
	Json(const char* s) : m_type(String), m_str(s) {}
	static Json array() { Json j; j.m_type = Array; return j; }
	static Json object() { Json j; j.m_type = Object; return j; }
	static Json binary(const void* data, size_t n_bytes) {
		Json j; j.m_type = Binary;
		j.m_bin = std::make_shared<std::vector<uint8_t>>((const uint8_t*)data, (const uint8_t*)data + n_bytes);
		return j;
	}
	static Json integer(long long v) { return Json(v); }

	// MessagePack (the subset nlohmann::json emits: nil, bool, ints, float32/64, str, bin, array, map with string keys)
	std::string to_msgpack() const;
	static Json from_msgpack(const void* data, size_t n_bytes);
	static Json from_msgpack_file(const std::string& path);
	void to_msgpack_file(const std::string& path) const;

	static Json parse(const std::string& text);
	static Json parse_file(const std::string& path);
	std::string dump() const;

	Type type() const { return m_type; }
	bool is_null() const { return m_type == Null; }
	bool is_object() const { return m_type == Object; }
	bool is_array() const { return m_type == Array; }
	bool is_number() const { return m_type == Number; }
	bool is_string() const { return m_type == String; }
	bool is_bool() const { return m_type == Bool; }
	bool is_binary() const { return m_type == Binary; }
	bool is_integer() const { return m_type == Number && m_is_int; }
	const std::vector<uint8_t>& bin() const { if (m_type != Binary) throw std::runtime_error("json: not a binary value"); return *m_bin; }

	bool contains(const std::string& key) const { return m_type == Object && m_obj.count(key) > 0; }
	Json& operator[](const std::string& key) { if (m_type == Null) m_type = Object; if (m_type != Object) throw std::runtime_error("json: not an object"); return m_obj[key]; }
	const Json& at(const std::string& key) const { auto it = m_obj.find(key); if (m_type != Object || it == m_obj.end()) throw std::runtime_error("json: missing key '" + key + "'"); return it->second; }
	const Json& operator[](const std::string& key) const { return at(key); }
	Json& operator[](size_t i) { return m_arr.at(i); }
	const Json& operator[](size_t i) const { return m_arr.at(i); }
	size_t size() const { return m_type == Array ? m_arr.size() : m_obj.size(); }
	void push_back(const Json& v) { if (m_type == Null) m_type = Array; m_arr.push_back(v); }
	void erase(const std::string& key) { m_obj.erase(key); }
	const std::map<std::string, Json>& items() const { return m_obj; }
	const std::vector<Json>& elements() const { return m_arr; }

	double number() const { if (m_type == Bool) return m_bool ? 1.0 : 0.0; if (m_type != Number) throw std::runtime_error("json: not a number"); return m_num; }
	const std::string& str() const { if (m_type != String) throw std::runtime_error("json: not a string"); return m_str; }
	bool boolean() const { if (m_type == Number) return m_num != 0.0; if (m_type != Bool) throw std::runtime_error("json: not a bool"); return m_bool; }

	double value(const std::string& key, double def) const { return contains(key) ? at(key).number() : def; }
	int value(const std::string& key, int def) const { return contains(key) ? (int)at(key).number() : def; }
	bool value(const std::string& key, bool def) const { return contains(key) ? at(key).boolean() : def; }
	std::string value(const std::string& key, const char* def) const { return contains(key) ? at(key).str() : std::string(def); }

	// RFC 7386
	void merge_patch(const Json& patch);

private:
	Type m_type;
	bool m_bool = false;
	double m_num = 0.0;
	bool m_is_int = false;   // written as a MessagePack integer / parsed from an integer literal
	std::shared_ptr<std::vector<uint8_t>> m_bin;
	std::string m_str;
	std::vector<Json> m_arr;
	std::map<std::string, Json> m_obj;
	friend class JsonParser;
};

} // namespace ngp
