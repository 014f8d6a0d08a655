"""genomicsdb_amd - MI355X-native variant-combine engine (scan + BroadCombinedGVCF hot path of GenomicsDB).

Python here is plumbing over the C ABI in include/genomicsdb_amd.h (libgenomicsdb_amd.so); the work happens in
the HIP kernels.  Mirrors the reference's caller-facing objects:
  GenomicsDBQueryStream  ~ com.intel.genomicsdb.reader.GenomicsDBQueryStream (JNI stream over GenomicsDBBCFGenerator)
  CombineEngine          ~ VariantQueryProcessor::scan_and_operate + BroadCombinedGVCFOperator on one column partition
"""
from .api import CombineEngine, GenomicsDBQueryStream, GenomicsDBException, import_cells, bgzf_compress, BGZF_EOF  # noqa: F401
