extern "C" const char* uz_source_hash(void) { return "a0917e9e697231b6b543bc2f896cb2b73510725be4ebf07dba0c978ad31eabeb"; }
