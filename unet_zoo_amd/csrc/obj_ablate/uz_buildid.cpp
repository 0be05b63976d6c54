extern "C" const char* uz_source_hash(void) { return "4351658698f7590e6fbf376fc36b865cc03b36a04f80023a82a6792e8efae26b"; }
