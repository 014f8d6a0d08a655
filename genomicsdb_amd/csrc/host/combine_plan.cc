#include "combine_plan.h"

#include <algorithm>
#include <array>
#include <map>
#include <unordered_map>
#include <cstring>
#include <set>

namespace genomicsdb_amd {

std::string HostPlan::bcf_header_bytes(bool keep_idx_fields) const {
  std::string text;
  for (size_t i = 0; i < header_lines.size(); ++i) {
    const std::string& l = header_lines[i];
    if (keep_idx_fields && header_line_idx[i] >= 0 && !l.empty() && l.back() == '>') text += l.substr(0, l.size() - 1) + ",IDX=" + std::to_string(header_line_idx[i]) + ">";
    else text += l;
    text += '\n';
  }
  text += header_text.substr(header_text.rfind("#CHROM"));
  std::string out("BCF\2\2", 5);
  const uint32_t l_text = (uint32_t)text.size() + 1;
  out.append((const char*)&l_text, 4);
  out += text;
  out.push_back('\0');
  return out;
}

HostPlan build_combine_plan(const VariantQueryConfig& qc, const std::string& tmpl, const std::string& output_format, bool use_missing_values_not_vector_end) {
  if (!qc.is_bookkeeping_done()) throw BroadCombinedGVCFException("do_query_bookkeeping() must run first");
  const VidMapper& vid = qc.get_vid_mapper();
  HostPlan hp;
  CombinePlan& pl = hp.plan;
  memset(&pl, 0, sizeof(pl));
  pl.f_REF = pl.f_ALT = pl.f_GT = pl.f_PL = pl.f_DP = pl.f_MIN_DP = pl.f_DP_FORMAT = pl.f_FILTER = pl.f_QUAL = pl.f_ID = -1;
  pl.qual_combine_op = GDB_OP_UNKNOWN;
  // ---- plan fields = queried attributes minus END -------------------------------------------------
  std::vector<int> q2f(qc.get_num_queried_attributes(), -1);
  for (unsigned q = 0; q < qc.get_num_queried_attributes(); ++q) {
    const std::string& name = qc.get_query_attribute_name(q);
    if (name == "END") continue;
    const FieldInfo* fi = qc.get_field_info_for_query_attribute_idx(q);
    if (fi->m_unsupported_on_device) throw UnsupportedOnDeviceException("field " + name + " has more than 2 dimensions");
    if (fi->m_num_dimensions == 2 && !(fi->m_is_vcf_INFO_field && !fi->m_is_vcf_FORMAT_field))
      throw UnsupportedOnDeviceException("2-dimensional FORMAT field " + name + ": only INFO annotations are combined on the device");
    if (pl.nfields >= GDB_MAX_FIELDS) throw UnsupportedOnDeviceException("more than GDB_MAX_FIELDS queried attributes");
    int f = pl.nfields++;
    q2f[q] = f;
    hp.field_names.push_back(name);
    GdbFieldDesc& d = pl.field[f];
    d.elem = fi->m_num_dimensions == 2 ? GDB_ET_CHAR : fi->m_element_type;   // a 2-D field is a byte blob (gdb_asa.hpp)
    d.ndim = (int32_t)fi->m_num_dimensions;
    d.elem2d = fi->m_element_type;
    d.delim0 = (unsigned char)fi->m_vcf_delimiter[0];
    d.delim1 = (unsigned char)fi->m_vcf_delimiter[1];
    d.length = fi->m_length_descriptor;
    d.fixed_num = (int)fi->m_num_elements;
    d.combine_op = fi->m_VCF_field_combine_operation;
    d.is_info = fi->m_is_vcf_INFO_field;
    d.is_format = fi->m_is_vcf_FORMAT_field;
    d.known_enum = qc.get_known_field_enum_for_query_idx(q);
    switch (d.known_enum) {
      case GVCF_REF_IDX: pl.f_REF = f; break;
      case GVCF_ALT_IDX: pl.f_ALT = f; break;
      case GVCF_GT_IDX: pl.f_GT = f; break;
      case GVCF_PL_IDX: pl.f_PL = f; break;
      case GVCF_DP_IDX: pl.f_DP = f; break;
      case GVCF_MIN_DP_IDX: pl.f_MIN_DP = f; break;
      case GVCF_DP_FORMAT_IDX: pl.f_DP_FORMAT = f; break;
      case GVCF_FILTER_IDX: pl.f_FILTER = f; break;
      case GVCF_QUAL_IDX: pl.f_QUAL = f; break;
      case GVCF_ID_IDX: pl.f_ID = f; break;
      default: break;
    }
  }
  if (pl.f_REF < 0 || pl.f_ALT < 0) throw BroadCombinedGVCFException("REF and ALT must be part of the query");
  if (pl.f_GT >= 0 && !(pl.field[pl.f_GT].length == GDB_VL_P || pl.field[pl.f_GT].length == GDB_VL_PP))
    throw BroadCombinedGVCFException("GT must have length descriptor P or PP");
  // ---- header lines -----------------------------------------------------------------------------------
  std::vector<std::string> lines;
  for (size_t p = 0; p < tmpl.size();) {
    size_t e = tmpl.find('\n', p);
    if (e == std::string::npos) e = tmpl.size();
    if (e - p >= 2 && tmpl[p] == '#' && tmpl[p + 1] == '#') lines.push_back(tmpl.substr(p, e - p));
    p = e + 1;
  }
  if (lines.empty()) {  // VCFAdapter::initialize_default_header when no template is given
    lines = {"##fileformat=VCFv4.2", "##FILTER=<ID=PASS,Description=\"All filters passed\">",
             "##ALT=<ID=NON_REF,Description=\"Represents any possible alternative allele at this location\">",
             "##INFO=<ID=END,Number=1,Type=Integer,Description=\"Stop position of the interval\">"};
  }
  std::set<std::string> have[3], hdr_contigs, hdr_ids;  // [FILTER, INFO, FORMAT]
  for (auto& l : lines) {
    int cls = l.rfind("##FILTER=", 0) == 0 ? 0 : l.rfind("##INFO=", 0) == 0 ? 1 : l.rfind("##FORMAT=", 0) == 0 ? 2 : -1;
    size_t idp = l.find("<ID=");
    if (idp == std::string::npos) continue;
    size_t ide = l.find_first_of(",>", idp + 4);
    std::string id = l.substr(idp + 4, ide - idp - 4);
    if (cls >= 0) { have[cls].insert(id); hdr_ids.insert(id); }
    if (l.rfind("##contig=", 0) == 0) hdr_contigs.insert(id);
  }
  auto add_field_to_hdr_if_missing = [&](const std::string& name, int cls) {
    const FieldInfo* fi = vid.get_field_info(name);
    // a multi-D or tuple field is a String with Number=1 in the header: a template line that says otherwise is removed and
    // added again, keeping its description (vcf_adapter.cc:62-95)
    const bool multid = fi && cls != 0 && (fi->get_num_elements_in_tuple() > 1u || fi->m_num_dimensions > 1u);
    std::string kept_description;
    if (multid && have[cls].count(name)) {
      const std::string prefix = std::string("##") + (cls == 1 ? "INFO" : "FORMAT") + "=<ID=" + name + ",";
      for (size_t li = 0; li < lines.size(); ++li)
        if (lines[li].rfind(prefix, 0) == 0) {
          const size_t dp = lines[li].find("Description="), de = lines[li].rfind('>');
          if (dp != std::string::npos && de != std::string::npos && de > dp + 12) kept_description = lines[li].substr(dp + 12, de - dp - 12);
          lines.erase(lines.begin() + (long)li);
          break;
        }
      have[cls].erase(name);
    }
    if (have[cls].count(name)) return;
    std::string h = std::string("##") + (cls == 0 ? "FILTER" : cls == 1 ? "INFO" : "FORMAT") + "=<ID=" + name;
    if (multid) {
      h += ",Number=1,Type=String,Description=" + (kept_description.empty() ? "\"" + name + "\"" : kept_description);
    } else if (cls == 2 && name == "GT") h += ",Number=1,Type=String,Description=\"Genotype\"";
    else {
      if (cls != 0) {
        if (!fi) throw BroadCombinedGVCFException("no vid info for header field " + name);
        h += ",Number=";
        if (fi->m_element_type == GDB_ET_FLAG) h += "0";
        else switch (fi->m_length_descriptor) {
          case GDB_VL_FIXED: h += std::to_string(fi->m_num_elements); break;
          case GDB_VL_VAR: h += "."; break;
          case GDB_VL_A: h += "A"; break;
          case GDB_VL_R: h += "R"; break;
          case GDB_VL_G: h += "G"; break;
          default: throw BroadCombinedGVCFException("Unhandled field length descriptor for " + name);
        }
        h += ",Type=";
        h += fi->m_element_type == GDB_ET_FLAG ? "Flag" : fi->m_element_type == GDB_ET_INT ? "Integer" : fi->m_element_type == GDB_ET_FLOAT ? "Float" : "String";
      }
      h += ",Description=\"" + name + "\"";
    }
    h += ">";
    lines.push_back(h);
    have[cls].insert(name);
    hdr_ids.insert(name);
  };
  // ---- INFO / FORMAT emission lists (broad_combined_gvcf.cc:163-263) ----------------------------------
  const bool sites_only = qc.sites_only_query();
  int dp_info_plan_field = -1;
  std::vector<int> fmt_list;
  // composite vid field idx -> plan fields of (bins, counts).  The reference keeps these pairs in a std::unordered_map<unsigned, ...>
  // filled in query order (broad_combined_gvcf.h:123, .cc:213-219) and emits the histogram_sum fields in that map's ITERATION order
  // (.cc:559): the order of two or more such fields is the C++ library's (libstdc++: a key whose bucket is empty goes to the head of
  // the list, so with few fields the one queried last comes out first).  Same container, same inserts, same order.
  std::unordered_map<unsigned, std::array<int, 2>> histogram_pairs;
  for (unsigned q = 0; q < qc.get_num_queried_attributes(); ++q) {
    int f = q2f[q];
    if (f < 0) continue;
    const FieldInfo* fi = qc.get_field_info_for_query_attribute_idx(q);
    int ke = qc.get_known_field_enum_for_query_idx(q);
    GdbCombineOp op = fi->m_VCF_field_combine_operation;
    bool add_INFO = fi->m_is_vcf_INFO_field && ke != GVCF_END_IDX && (ke != GVCF_DP_IDX || op != GDB_OP_DP) && op != GDB_OP_MOVE_TO_FORMAT;
    bool add_FORMAT = (fi->m_is_vcf_FORMAT_field && (!sites_only || ke == GVCF_DP_FORMAT_IDX || ke == GVCF_MIN_DP_IDX)) ||
                      (fi->m_is_vcf_INFO_field && ((ke == GVCF_DP_IDX && op == GDB_OP_DP) || (op == GDB_OP_MOVE_TO_FORMAT && !sites_only)));
    if (add_INFO && op == GDB_OP_HISTOGRAM_SUM) {   // the (bins, counts) pair of a composite field (broad_combined_gvcf.cc:194-221)
      add_field_to_hdr_if_missing(fi->m_vcf_name, 1);
      if (!fi->is_flattened_field()) throw BroadCombinedGVCFException("Operation histogram_sum needs a field whose elements are tuples; field " + fi->m_name);
      const FieldInfo& parent = vid.get_field_info((unsigned)fi->m_parent_composite_field_idx);
      if (parent.get_num_elements_in_tuple() != 2u)
        throw BroadCombinedGVCFException("Operation histogram_sum is only supported for fields whose elements are tuple with 2 constituent elements; field " + parent.m_name +
                                         " does not satisfy this requirement");
      if (fi->m_element_type != GDB_ET_INT && fi->m_element_type != GDB_ET_FLOAT)
        throw BroadCombinedGVCFException("Operation histogram_sum is only supported for tuple elements that are int or float; field " + parent.m_name);
      if (fi->m_num_dimensions != 2) throw UnsupportedOnDeviceException("histogram_sum over a field that is not 2-dimensional: " + parent.m_name);
      auto it = histogram_pairs.insert(std::make_pair((unsigned)fi->m_parent_composite_field_idx, std::array<int, 2>{{-1, -1}})).first;
      it->second[fi->m_element_index_in_tuple == 0u ? 0 : 1] = f;
      continue;
    }
    if (add_INFO && op != GDB_OP_UNKNOWN) {  // UNKNOWN: "field will NOT be part of INFO fields" warning in the reference
      if (op != GDB_OP_SUM && op != GDB_OP_MEAN && op != GDB_OP_MEDIAN && op != GDB_OP_ELEMENT_WISE_SUM && op != GDB_OP_CONCATENATE)
        throw UnsupportedOnDeviceException("INFO combine operation of field " + fi->m_name + " is not on the device path");
      if (fi->m_num_dimensions == 2 && op != GDB_OP_ELEMENT_WISE_SUM)
        throw UnsupportedOnDeviceException("2-dimensional INFO field " + fi->m_name + ": only element_wise_sum and histogram_sum are defined for it");
      if (fi->m_element_type != GDB_ET_INT && fi->m_element_type != GDB_ET_FLOAT)
        throw UnsupportedOnDeviceException("INFO reducer on non-numeric field " + fi->m_name);
      if (pl.n_info >= GDB_MAX_INFO_FIELDS) throw UnsupportedOnDeviceException("too many INFO fields");
      pl.info_field[pl.n_info++] = f;
      add_field_to_hdr_if_missing(fi->m_vcf_name, 1);
    }
    if (add_FORMAT) {
      if (fi->m_is_vcf_FORMAT_field || op == GDB_OP_MOVE_TO_FORMAT) {
        if (fi->is_length_allele_dependent() && fi->m_element_type != GDB_ET_INT && fi->m_element_type != GDB_ET_FLOAT)
          throw UnsupportedOnDeviceException("allele-dependent non-numeric FORMAT field " + fi->m_name);
        fmt_list.push_back(f);
        add_field_to_hdr_if_missing(fi->m_vcf_name, 2);
      } else {  // INFO DP handled with the FORMAT fields
        dp_info_plan_field = f;
        add_field_to_hdr_if_missing("DP", 1);
      }
    }
  }
  for (auto& kv : histogram_pairs) {
    if (pl.n_histogram >= GDB_MAX_HISTOGRAM_FIELDS) throw UnsupportedOnDeviceException("too many histogram_sum fields");
    if (kv.second[0] < 0 || kv.second[1] < 0)
      throw BroadCombinedGVCFException("histogram_sum needs both tuple elements of field " + vid.get_field_info(kv.first).m_name + " among the queried attributes");
    pl.histogram_bin_field[pl.n_histogram] = kv.second[0];
    pl.histogram_count_field[pl.n_histogram] = kv.second[1];
    ++pl.n_histogram;
  }
  if (pl.f_FILTER >= 0)
    for (unsigned i = 0; i < vid.get_num_fields(); ++i) if (vid.get_field_info(i).m_is_vcf_FILTER_field) add_field_to_hdr_if_missing(vid.get_field_info(i).m_vcf_name, 0);
  // emission order on the device: GT first (htslib bcf_update_format moves GT to the front), DP_FORMAT is never
  // inserted by itself, the DP pseudo entry (values of DP_FORMAT) goes last (broad_combined_gvcf.cc:667-719)
  if (!sites_only) {
    for (int f : fmt_list) if (f == pl.f_GT) pl.format_field[pl.n_format++] = f;
    for (int f : fmt_list) {
      if (f == pl.f_GT || f == pl.f_DP_FORMAT) continue;
      if (pl.n_format >= GDB_MAX_FORMAT_FIELDS - 1) throw UnsupportedOnDeviceException("too many FORMAT fields");
      pl.format_field[pl.n_format++] = f;
    }
    if (pl.f_DP_FORMAT >= 0 && (dp_info_plan_field >= 0 || std::find(fmt_list.begin(), fmt_list.end(), pl.f_DP_FORMAT) != fmt_list.end()))
      pl.format_field[pl.n_format++] = dp_info_plan_field >= 0 ? dp_info_plan_field : pl.f_DP_FORMAT;
  }
  if (dp_info_plan_field < 0 && pl.f_DP >= 0) {
    // DP queried but with a user combine op: it is an ordinary INFO field then; the DP-sum rule is off
    pl.f_DP = -1;
  }
  const FieldInfo* qual_info = vid.get_field_info("QUAL");
  if (qual_info && pl.f_QUAL >= 0 && qual_info->m_VCF_field_combine_operation != GDB_OP_UNKNOWN) {
    GdbCombineOp op = qual_info->m_VCF_field_combine_operation;
    if (op != GDB_OP_SUM && op != GDB_OP_MEAN && op != GDB_OP_MEDIAN) throw UnsupportedOnDeviceException("QUAL combine operation");
    pl.qual_combine_op = op;
  }
  pl.produce_GT_field = qc.produce_GT_field();
  pl.produce_FILTER_field = qc.produce_FILTER_field() && pl.f_FILTER >= 0;
  pl.sites_only_query = sites_only;
  pl.min_PL_GT_for_spanning_deletions = qc.produce_GT_with_min_PL_value_for_spanning_deletions();
  pl.max_diploid_alt_alleles = (int)qc.get_max_diploid_alt_alleles_that_can_be_genotyped();
  pl.id_order_unordered_set = qc.id_union_order_unordered_set() ? 1 : 0;
  pl.num_query_rows = (int)qc.get_num_rows_to_query();
  // ---- contigs ---------------------------------------------------------------------------------------
  for (unsigned i = 0; i < vid.get_num_contigs(); ++i) {
    const ContigInfo& c = vid.get_contig_info(i);
    if (!hdr_contigs.count(c.m_name)) lines.push_back("##contig=<ID=" + c.m_name + ",length=" + std::to_string(c.m_length) + ">");
    GdbContig g;
    g.offset = c.m_tiledb_column_offset;
    g.length = c.m_length;
    g.name_off = (int32_t)hp.contig_names.size();
    g.name_len = (int32_t)c.m_name.size();
    hp.contig_names += c.m_name;
    hp.contigs.push_back(g);
  }
  // ---- header dictionaries (htslib bcf_hdr_parse): FILTER / INFO / FORMAT ids share one dictionary, PASS is entry 0, the others
  // follow in order of first appearance; contigs have their own, in line order
  {
    std::vector<std::pair<std::string, int>> dict;   // id -> index
    auto find_id = [&](const std::string& id) -> int { for (auto& d : dict) if (d.first == id) return d.second; return -1; };
    dict.push_back(std::make_pair(std::string("PASS"), 0));
    std::vector<std::pair<std::string, int>> ctg_dict;
    for (auto& l : lines) {
      int idx = -1;
      const bool is_dict = l.rfind("##FILTER=", 0) == 0 || l.rfind("##INFO=", 0) == 0 || l.rfind("##FORMAT=", 0) == 0;
      const bool is_ctg = l.rfind("##contig=", 0) == 0;
      size_t idp = l.find("<ID=");
      if ((is_dict || is_ctg) && idp != std::string::npos) {
        size_t ide = l.find_first_of(",>", idp + 4);
        const std::string id = l.substr(idp + 4, ide - idp - 4);
        if (is_dict) { idx = find_id(id); if (idx < 0) { idx = (int)dict.size(); dict.push_back(std::make_pair(id, idx)); } }
        else { idx = (int)ctg_dict.size(); ctg_dict.push_back(std::make_pair(id, idx)); }
      }
      hp.header_lines.push_back(l);
      hp.header_line_idx.push_back(idx);
    }
    for (int f = 0; f < GDB_MAX_FIELDS; ++f) pl.bcf_id[f] = -1;
    for (int f = 0; f < pl.nfields; ++f) { const FieldInfo* fi = vid.get_field_info(hp.field_names[(size_t)f]); if (fi) pl.bcf_id[f] = find_id(fi->m_vcf_name); }
    pl.bcf_end_id = find_id("END");
    pl.bcf_dp_id = find_id("DP");
    for (unsigned i = 0; i < vid.get_num_fields(); ++i) {
      const FieldInfo& fi = vid.get_field_info(i);
      hp.filter_bcf_id.push_back(have[0].count(fi.m_vcf_name) ? find_id(fi.m_vcf_name) : -1);
    }
    for (auto& g : hp.contigs) {
      const std::string name = hp.contig_names.substr((size_t)g.name_off, (size_t)g.name_len);
      g.rid = -1; g.pad = 0;
      for (auto& c : ctg_dict) if (c.first == name) g.rid = c.second;
    }
  }
  // "": VCF text, "bu": uncompressed BCF2, "z": BGZF-compressed VCF text, "b": BGZF-compressed BCF2 (the modes htslib's hts_open
  // takes behind "w": vcf_adapter.cc:358-363); "v" is htslib's explicit spelling of uncompressed text
  pl.bcf_mode = (output_format == "bu" || output_format == "b") ? 1 : 0;
  hp.bgzf = output_format == "z" || output_format == "b";
  if (!(output_format.empty() || output_format == "v" || output_format == "bu" || output_format == "z" || output_format == "b"))
    throw UnsupportedOnDeviceException("VCF output format \"" + output_format + "\": known formats are \"\" (VCF text), \"z\" (BGZF-compressed VCF), \"bu\" (BCF2) and \"b\" (BGZF-compressed BCF2)");
  pl.use_missing_values_not_vector_end = use_missing_values_not_vector_end ? 1 : 0;
  pl.bcf_n_sample = sites_only ? 0 : (int32_t)qc.get_num_rows_to_query();
  if (pl.bcf_mode && pl.bcf_end_id < 0) throw BroadCombinedGVCFException("BCF output needs an INFO END line in the header");
  std::sort(hp.contigs.begin(), hp.contigs.end(), [](const GdbContig& a, const GdbContig& b) { return a.offset < b.offset; });
  for (auto& l : lines) { hp.header_text += l; hp.header_text += '\n'; }
  hp.header_text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
  if (!sites_only) {
    hp.header_text += "\tFORMAT";
    for (uint64_t i = 0; i < qc.get_num_rows_to_query(); ++i) {
      std::string nm;
      int64_t row = qc.get_array_row_idx_for_query_row_idx(i);
      if (!vid.get_callset_name(row, nm) || nm.empty())
        throw BroadCombinedGVCFException("No sample/CallSet name specified in JSON file/Protobuf object for TileDB row " + std::to_string(row));
      hp.header_text += '\t';
      hp.header_text += nm;
    }
  }
  hp.header_text += '\n';
  // ---- name tables ----------------------------------------------------------------------------------
  for (int f = 0; f < pl.nfields; ++f) {
    const FieldInfo* fi = vid.get_field_info(hp.field_names[f]);
    hp.field_name_off.push_back((int32_t)hp.names_text.size());
    hp.field_name_len.push_back((int32_t)fi->m_vcf_name.size());
    hp.names_text += fi->m_vcf_name;
  }
  for (unsigned i = 0; i < vid.get_num_fields(); ++i) {
    const FieldInfo& fi = vid.get_field_info(i);
    hp.filter_name_off.push_back((int32_t)hp.names_text.size());
    if (hdr_ids.count(fi.m_vcf_name) && have[0].count(fi.m_vcf_name)) { hp.filter_name_len.push_back((int32_t)fi.m_vcf_name.size()); hp.names_text += fi.m_vcf_name; }
    else hp.filter_name_len.push_back(0);
  }
  return hp;
}

}  // namespace genomicsdb_amd
