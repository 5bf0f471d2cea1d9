// ht_json.hpp -- the one JSON reader of the product (the subset the reference's files use: model files, dataset headers, tracker configuration).  Header-only, no
// exceptions: a parse error is a message in jparser::err.  Numbers keep their text; floats are converted with strtof, which is what the reference's
// `istringstream >> float` does (third_party/json.h:104), so every value is the correctly rounded fp32 of its decimal text.
// Used by csrc/ht_model_build.hip (model_hand.json, config.json) and include/ht_formats.hpp (the data sets' .json headers).
#pragma once
#include <string>
#include <vector>
#include <string.h>
#include <stdlib.h>

namespace ht_json {
struct jnode
{
	enum kind_t { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
	std::string text;
	std::vector<jnode> items;
	std::vector<std::string> keys;
	const jnode *get(const char *k) const { for (size_t i = 0; i < keys.size(); i++) if (keys[i] == k) return &items[i]; return nullptr; }
	float as_float() const { return kind == NUM ? strtof(text.c_str(), nullptr) : 0.0f; }
	int as_int() const { return kind == NUM ? (int)strtol(text.c_str(), nullptr, 10) : 0; }
};
struct jparser
{
	const char *p, *end; std::string err;
	void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
	bool fail(const char *m) { if (err.empty()) err = m; return false; }
	bool value(jnode &n, int depth)
	{
		if (depth > 64) return fail("json nesting too deep");
		ws();
		if (p >= end) return fail("unexpected end of json");
		if (*p == '{')
		{
			n.kind = jnode::OBJ; p++; ws();
			if (p < end && *p == '}') { p++; return true; }
			for (;;)
			{
				jnode k; ws();
				if (p >= end || *p != '"' || !string(k)) return fail("object key expected");
				ws(); if (p >= end || *p != ':') return fail("':' expected"); p++;
				n.keys.push_back(k.text); n.items.emplace_back();
				if (!value(n.items.back(), depth + 1)) return false;
				ws(); if (p < end && *p == ',') { p++; continue; }
				if (p < end && *p == '}') { p++; return true; }
				return fail("',' or '}' expected");
			}
		}
		if (*p == '[')
		{
			n.kind = jnode::ARR; p++; ws();
			if (p < end && *p == ']') { p++; return true; }
			for (;;)
			{
				n.items.emplace_back();
				if (!value(n.items.back(), depth + 1)) return false;
				ws(); if (p < end && *p == ',') { p++; continue; }
				if (p < end && *p == ']') { p++; return true; }
				return fail("',' or ']' expected");
			}
		}
		if (*p == '"') return string(n);
		if (!strncmp(p, "true", 4) && end - p >= 4) { n.kind = jnode::BOOL; n.text = "1"; p += 4; return true; }
		if (!strncmp(p, "false", 5) && end - p >= 5) { n.kind = jnode::BOOL; n.text = "0"; p += 5; return true; }
		if (!strncmp(p, "null", 4) && end - p >= 4) { n.kind = jnode::NUL; p += 4; return true; }
		const char *s = p;
		while (p < end && (strchr("+-.eE", *p) || (*p >= '0' && *p <= '9'))) p++;
		if (p == s) return fail("unexpected character in json");
		n.kind = jnode::NUM; n.text.assign(s, p);
		return true;
	}
	bool string(jnode &n)
	{
		n.kind = jnode::STR; p++;
		while (p < end && *p != '"') { if (*p == '\\' && p + 1 < end) p++; n.text.push_back(*p++); }
		if (p >= end) return fail("unterminated string");
		p++; return true;
	}
};
// parses a whole document; false with *err set on a syntax error
inline bool parse(const char *text, size_t n, jnode &root, std::string *err = nullptr)
{
	jparser jp = { text, text + n, "" };
	if (jp.value(root, 0)) return true;
	if (err) *err = jp.err;
	return false;
}
}  // namespace ht_json
