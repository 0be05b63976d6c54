extern "C" const char* uz_source_hash(void) { return "b6e587b5cdd2cb60df16bc4ece022c8925ebbe32c55982d7e29c2a667cc2c6c5"; }
