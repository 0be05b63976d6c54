extern "C" const char* uz_source_hash(void) { return "6c37d7b7ee6a980381867d2fdcc904f51e84ee1a8a13be322f2f04ba414bdb39"; }
