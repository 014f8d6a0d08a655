// vcf_index.h - .tbi / .csi next to a BGZF-compressed output file ("index_output_VCF": true).
//
// The reference asks htslib for it when its file-writing VCFAdapter closes (src/main/cpp/src/vcf/vcf_adapter.cc:275-295:
// tbx_index_build(file, 0, &tbx_conf_vcf) for "z", bcf_index_build(file, 14) for "b"); both htslib functions re-read the finished file.
// htslib is not in the tree (empty submodule): this is a restatement of the two published formats - the tabix index ("TBI\1", The Tabix
// index file format) and the coordinate-sorted index ("CSI\1", CSIv1) - with the UCSC binning scheme both use (min_shift 14, depth 5).
// Host code by nature (the reference's is, too): it walks the BGZF blocks of the file that was written, inflates them with zlib and notes, per
// record, contig, interval and the virtual offsets (block start << 16 | offset inside the block) in front of and behind it.
#pragma once
#include <string>

namespace genomicsdb_amd {

// <path>.tbi for a bgzip'ed VCF (records indexed by CHROM, POS and INFO END= / the REF allele's length, as tabix does for its VCF preset)
void build_tbi_index(const std::string& vcf_gz_path);
// <path>.csi for a bgzip'ed BCF2 file (CHROM index, POS, rlen of the records; min_shift 14 like the reference's call)
void build_csi_index(const std::string& bcf_path);

}  // namespace genomicsdb_amd
