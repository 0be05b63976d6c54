extern "C" const char* uz_source_hash(void) { return "702bd7f0a8a9f4c1a66f5aa50fba74f0bb5929efd93ac3895d7370ebc2f26ea1"; }
