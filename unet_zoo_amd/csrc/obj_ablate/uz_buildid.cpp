extern "C" const char* uz_source_hash(void) { return "0337b6f8c3f3a2c4c08709e7abdaf5f1b09f7da1a007676575ea6b4bdfc020c2"; }
