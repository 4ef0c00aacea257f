// WordPiece tokenizer (BERT "cased" flavour) in C++: the role the Rust `tokenizers` library plays for the
// reference (AutoTokenizer at /root/reference/test_data_parallelism.py:69,75).  Basic tokenisation
// (whitespace / punctuation / CJK splitting, control-character cleanup, optional lower-casing of ASCII and
// Latin-1) followed by greedy longest-match-first WordPiece with "##" continuations; sentence pairs are
// assembled as [CLS] a [SEP] b [SEP] with token types 0/1 and "longest first" truncation.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace py = pybind11;

namespace {

std::vector<uint32_t> decode_utf8(const std::string& s) {
  std::vector<uint32_t> out;
  out.reserve(s.size());
  size_t i = 0, n = s.size();
  while (i < n) {
    unsigned char c = (unsigned char)s[i];
    uint32_t cp;
    int len;
    if (c < 0x80) { cp = c; len = 1; }
    else if ((c >> 5) == 0x6) { cp = c & 0x1F; len = 2; }
    else if ((c >> 4) == 0xE) { cp = c & 0x0F; len = 3; }
    else if ((c >> 3) == 0x1E) { cp = c & 0x07; len = 4; }
    else { cp = 0xFFFD; len = 1; }
    if (i + len > n) { out.push_back(0xFFFD); break; }
    for (int k = 1; k < len; ++k) cp = (cp << 6) | ((unsigned char)s[i + k] & 0x3F);
    out.push_back(cp);
    i += len;
  }
  return out;
}

void append_utf8(std::string& s, uint32_t cp) {
  if (cp < 0x80) s.push_back((char)cp);
  else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

bool is_whitespace(uint32_t c) {
  return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
         c == 0x202F || c == 0x205F || c == 0x3000;
}
bool is_control(uint32_t c) {
  if (c == '\t' || c == '\n' || c == '\r') return false;
  return c < 0x20 || (c >= 0x7F && c < 0xA0) || c == 0xAD || (c >= 0x200B && c <= 0x200F) || (c >= 0x202A && c <= 0x202E) ||
         (c >= 0x2060 && c <= 0x2064) || c == 0xFEFF;
}
bool is_punct(uint32_t c) {
  if ((c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126)) return true;
  // Unicode punctuation blocks that occur in news text (general punctuation, CJK symbols, Latin-1 marks)
  return c == 0xA1 || c == 0xA7 || c == 0xAB || c == 0xB6 || c == 0xB7 || c == 0xBB || c == 0xBF ||
         (c >= 0x2010 && c <= 0x2027) || (c >= 0x2030 && c <= 0x205E) || (c >= 0x3001 && c <= 0x3003) ||
         (c >= 0x3008 && c <= 0x3011) || (c >= 0x3014 && c <= 0x301F) || (c >= 0xFF01 && c <= 0xFF0F) ||
         (c >= 0xFF1A && c <= 0xFF20) || (c >= 0xFF3B && c <= 0xFF40) || (c >= 0xFF5B && c <= 0xFF65);
}
bool is_cjk(uint32_t c) {
  return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF) || (c >= 0x20000 && c <= 0x2A6DF) ||
         (c >= 0x2A700 && c <= 0x2B73F) || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B820 && c <= 0x2CEAF) ||
         (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}
uint32_t lower(uint32_t c) {
  if (c >= 'A' && c <= 'Z') return c + 32;
  if (c >= 0xC0 && c <= 0xDE && c != 0xD7) return c + 32;
  return c;
}

class WordPiece {
 public:
  WordPiece(const std::string& vocab_file, bool do_lower_case, int max_chars_per_word)
      : lower_(do_lower_case), max_chars_(max_chars_per_word) {
    std::ifstream f(vocab_file);
    if (!f) throw std::runtime_error("cannot open vocabulary file " + vocab_file);
    std::string line;
    int64_t id = 0;
    while (std::getline(f, line)) {
      while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
      vocab_.emplace(line, id++);
      tokens_.push_back(line);
    }
    unk_ = must("[UNK]"); cls_ = must("[CLS]"); sep_ = must("[SEP]"); pad_ = must("[PAD]");
  }

  int64_t vocab_size() const { return (int64_t)tokens_.size(); }
  int64_t pad_id() const { return pad_; }
  int64_t cls_id() const { return cls_; }
  int64_t sep_id() const { return sep_; }
  std::string id_to_token(int64_t i) const { return tokens_.at((size_t)i); }

  std::vector<std::string> basic(const std::string& text) const {
    std::vector<std::string> words;
    std::string cur;
    auto flush = [&]() { if (!cur.empty()) { words.push_back(cur); cur.clear(); } };
    for (uint32_t c : decode_utf8(text)) {
      if (c == 0 || c == 0xFFFD || is_control(c)) continue;
      if (is_whitespace(c)) { flush(); continue; }
      if (lower_) c = lower(c);
      if (is_punct(c) || is_cjk(c)) {
        flush();
        std::string one;
        append_utf8(one, c);
        words.push_back(one);
      } else {
        append_utf8(cur, c);
      }
    }
    flush();
    return words;
  }

  void wordpiece(const std::string& word, std::vector<int64_t>& out) const {
    const std::vector<uint32_t> cps = decode_utf8(word);
    if ((int)cps.size() > max_chars_) { out.push_back(unk_); return; }
    std::vector<size_t> offs(cps.size() + 1, 0);  // byte offsets of code points
    {
      std::string tmp;
      for (size_t i = 0; i < cps.size(); ++i) { offs[i] = tmp.size(); append_utf8(tmp, cps[i]); }
      offs[cps.size()] = tmp.size();
    }
    std::vector<int64_t> pieces;
    size_t start = 0;
    while (start < cps.size()) {
      size_t end = cps.size();
      int64_t found = -1;
      while (start < end) {
        std::string sub = word.substr(offs[start], offs[end] - offs[start]);
        if (start > 0) sub = "##" + sub;
        auto it = vocab_.find(sub);
        if (it != vocab_.end()) { found = it->second; break; }
        --end;
      }
      if (found < 0) { out.push_back(unk_); return; }
      pieces.push_back(found);
      start = end;
    }
    out.insert(out.end(), pieces.begin(), pieces.end());
  }

  std::vector<int64_t> tokenize_ids(const std::string& text) const {
    std::vector<int64_t> ids;
    for (const auto& w : basic(text)) wordpiece(w, ids);
    return ids;
  }
  std::vector<std::string> tokenize(const std::string& text) const {
    std::vector<std::string> out;
    for (int64_t i : tokenize_ids(text)) out.push_back(tokens_[(size_t)i]);
    return out;
  }

  // [CLS] a [SEP] (b [SEP]) with "longest first" truncation to max_length; returns (input_ids, token_type_ids)
  std::pair<std::vector<int64_t>, std::vector<int64_t>> encode(const std::string& a, const std::string& b, bool has_b,
                                                                int max_length) const {
    std::vector<int64_t> ia = tokenize_ids(a), ib;
    if (has_b) ib = tokenize_ids(b);
    if (max_length > 0) {
      const int special = has_b ? 3 : 2;
      while ((int)(ia.size() + ib.size()) + special > max_length) {
        // one token at a time from the longer sequence; on a tie the second sequence gives way
        if (ia.size() > ib.size()) ia.pop_back();
        else if (!ib.empty()) ib.pop_back();
        else if (!ia.empty()) ia.pop_back();
        else break;
      }
    }
    std::vector<int64_t> ids, types;
    ids.push_back(cls_);
    ids.insert(ids.end(), ia.begin(), ia.end());
    ids.push_back(sep_);
    types.assign(ids.size(), 0);
    if (has_b) {
      ids.insert(ids.end(), ib.begin(), ib.end());
      ids.push_back(sep_);
      types.resize(ids.size(), 1);
    }
    return {ids, types};
  }

  py::list encode_pairs(const std::vector<std::string>& as, const std::vector<std::string>& bs, int max_length) const {
    if (!bs.empty() && bs.size() != as.size()) throw std::runtime_error("encode_pairs: list lengths differ");
    std::vector<std::pair<std::vector<int64_t>, std::vector<int64_t>>> enc(as.size());
    {
      py::gil_scoped_release rel;
      for (size_t i = 0; i < as.size(); ++i) enc[i] = encode(as[i], bs.empty() ? std::string() : bs[i], !bs.empty(), max_length);
    }
    py::list out;
    for (auto& e : enc) out.append(py::make_tuple(e.first, e.second));
    return out;
  }

 private:
  int64_t must(const char* tok) const {
    auto it = vocab_.find(tok);
    if (it == vocab_.end()) throw std::runtime_error(std::string("vocabulary lacks ") + tok);
    return it->second;
  }
  std::unordered_map<std::string, int64_t> vocab_;
  std::vector<std::string> tokens_;
  bool lower_;
  int max_chars_;
  int64_t unk_ = 0, cls_ = 0, sep_ = 0, pad_ = 0;
};

}  // namespace

void bind_wordpiece(py::module_& m) {
  py::class_<WordPiece>(m, "WordPiece")
      .def(py::init<const std::string&, bool, int>(), py::arg("vocab_file"), py::arg("do_lower_case") = false,
           py::arg("max_chars_per_word") = 100)
      .def("tokenize", &WordPiece::tokenize)
      .def("tokenize_ids", &WordPiece::tokenize_ids)
      .def("encode", [](const WordPiece& w, const std::string& a, py::object b, int max_length) {
             const bool has_b = !b.is_none();
             return w.encode(a, has_b ? b.cast<std::string>() : std::string(), has_b, max_length);
           }, py::arg("text"), py::arg("text_pair") = py::none(), py::arg("max_length") = 512)
      .def("encode_pairs", &WordPiece::encode_pairs, py::arg("texts"), py::arg("text_pairs") = std::vector<std::string>(),
           py::arg("max_length") = 512)
      .def("id_to_token", &WordPiece::id_to_token)
      .def_property_readonly("vocab_size", &WordPiece::vocab_size)
      .def_property_readonly("pad_token_id", &WordPiece::pad_id)
      .def_property_readonly("cls_token_id", &WordPiece::cls_id)
      .def_property_readonly("sep_token_id", &WordPiece::sep_id);
}
